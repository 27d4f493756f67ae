"""timing of nms_fused_kernel against the number of mask slices per class (debug flavour): python tools/nms_fused_slices.py [M] [n_cls]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multipathnet_amd
from multipathnet_amd import utils, _lib
from conftest import random_scored_boxes
lib = _lib.load("debug")
lib.mpn_debug_set_nms_fused(2)   # the fused kernel at every size it can take 
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for n_cls in ([int(sys.argv[2])] if len(sys.argv) > 2 else [1, 4, 20, 80]):
    rng = np.random.default_rng(0)
    d = torch.from_numpy(np.stack([random_scored_boxes(rng, M, "distinct") for _ in range(n_cls)])).to(dev)
    for S in (1, 2, 4, 8, 12, 16):
        lib.mpn_debug_set_nms_fused_slices(S)
        with _lib.debug_hooks():
            for _ in range(2):
                utils.nms_batched(d, None, 0.3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                utils.nms_batched(d, None, 0.3)
            e1.record(); torch.cuda.synchronize()
        print("M=%d, %2d classes, %2d slices (%3d blocks): %7.1f us/call" % (M, n_cls, S, min(S, (M + 31) // 32) * n_cls, e0.elapsed_time(e1) / 10 * 1e3))
lib.mpn_debug_set_nms_fused_slices(0)
