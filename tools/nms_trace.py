"""s_memtime trace of the chunked NMS scan (class 0): cycles per chunk spent in [row loads issued, diagonal resolved, boxes emitted,
rows folded, heads replayed]  (debug flavour of the library)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from multipathnet_amd import _lib, utils
from conftest import random_scored_boxes
lib = _lib.load("debug")
rng = np.random.default_rng(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for regime, mode in (("distinct", 0), ("dup", 0), ("fewties", 0), ("ties16", 0)):
    sb = np.stack([random_scored_boxes(rng, M, "distinct") for _ in range(4)])
    if regime == "fewties":
        for c in range(4):
            k = rng.choice(M, 8, replace=False)
            sb[c, k[:4], 4] = sb[c, k[4:], 4]
    if regime == "ties16":
        for c in range(4):
            k = rng.choice(M, 32, replace=False)
            sb[c, k[:16], 4] = sb[c, k[16:], 4]
    if regime == "dup":  # the bench image's regime: one duplicated proposal per class, high-scoring enough to be alive at its turn
        for c in range(4):
            o = np.argsort(-sb[c, :, 4])
            sb[c, o[5 + c]] = sb[c, o[40]] = sb[c, o[5 + c]].copy()
    d = torch.from_numpy(sb).cuda()
    buf = torch.zeros(64 * 8, dtype=torch.int64, device="cuda")
    with _lib.debug_hooks():
        utils.nms_batched(d, None, 0.3)
        lib.mpn_debug_set_nms_trace(C.c_void_p(buf.data_ptr()))
        keep, idx, nk = utils.nms_batched(d, None, 0.3)
        torch.cuda.synchronize()
        lib.mpn_debug_set_nms_trace(None)
    t = buf.cpu().view(64, 8).numpy()
    print(regime, "kept", nk.tolist(), " class 0: kernel %d cycles, of which %d in %d simulate() calls / %d batches" % tuple(t[63][4:8]))
    for c in range(min(16, (M + 63) // 64)):
        r = t[c]
        if r[0] == 0: continue
        print("  chunk %2d: start %8d  rows+diag %6d  resolve %6d  emit %6d  fold %6d  replay %6d" % (c, r[0] - t[0][0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]))
