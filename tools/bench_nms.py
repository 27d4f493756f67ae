"""micro-benchmark of mpn_nms_batched: per-path latency for the SURVEY §8d NMS micro-inputs"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import multipathnet_amd
from multipathnet_amd import utils
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import random_scored_boxes

lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n_cls, M = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for regime in ("distinct", "ties", "saturated"):
    sb = np.stack([random_scored_boxes(rng, M, regime) for _ in range(n_cls)])
    d = torch.from_numpy(sb).to(dev)
    for mode, name in ((0, "auto"), (2, "tie-kernel"), (1, "sweep-kernel")):
        lib.mpn_debug_set_nms_force_exact(mode)
        for _ in range(2):
            keep, idx, nk = utils.nms_batched(d, None, 0.3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            keep, idx, nk = utils.nms_batched(d, None, 0.3)
        e1.record()
        torch.cuda.synchronize()
        print("%-9s M=%d %-12s %8.1f us/call  kept/class mean %.0f" % (regime, M, name, e0.elapsed_time(e1) / 5 * 1e3, nk.float().mean().item()))
lib.mpn_debug_set_nms_force_exact(0)
