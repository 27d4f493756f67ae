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
only_regimes = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # e.g. fewties
only_modes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None   # e.g. 0,3
from multipathnet_amd import _lib
lib.mpn_debug_set_nms_fused_fence(int(os.environ.get("MPN_FUSED_FENCE", "1")))     # 0: round 5's relaxed block hand-off (no acquire-release counter)
lib.mpn_debug_set_nms_fused_replay(int(os.environ.get("MPN_FUSED_REPLAY", "1")))   # 0: the fused kernel's per-round tie path for every tied class
for regime in ("distinct", "fewties", "ties30", "ties100", "ties", "saturated"):
    if only_regimes and regime not in only_regimes:
        continue
    sb = np.stack([random_scored_boxes(rng, M, "distinct" if regime in ("fewties", "ties30", "ties100") else regime) for _ in range(n_cls)])
    if regime in ("ties30", "ties100"):  # 30 / 100 bit-equal pairs per class
        npairs = int(regime[4:])
        for c in range(n_cls):
            k = rng.choice(M, 2 * npairs, replace=False)
            sb[c, k[:npairs], 4] = sb[c, k[npairs:], 4]
    if regime == "fewties":  # what real softmax scores look like: a handful of bit-equal pairs per class
        for c in range(n_cls):
            k = rng.choice(M, 8, replace=False)
            sb[c, k[:4], 4] = sb[c, k[4:], 4]
    d = torch.from_numpy(sb).to(dev)
    for mode, name in ((0, "auto"), (5, "fused"), (4, "chain-auto"), (3, "replay-scan"), (2, "tie-kernel"), (1, "sweep-kernel")):
        if only_modes and mode not in only_modes:
            continue
        lib.mpn_debug_set_nms_force_exact(0 if mode == 5 else mode % 4)
        lib.mpn_debug_set_nms_fused(0 if mode == 4 else 2 if mode == 5 else 1)   # auto = the fused one-launch kernel for tables of <= 1024 rows (round 5); chain-auto = rounds 2-4
        with _lib.debug_hooks():
          for _ in range(2):
            keep, idx, nk = utils.nms_batched(d, None, 0.3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with _lib.debug_hooks():
          for _ in range(5):
            keep, idx, nk = utils.nms_batched(d, None, 0.3)
        e1.record()
        torch.cuda.synchronize()
        print("%-9s M=%d %-12s %8.1f us/call (20 classes, %.1f us/class if serial)  kept/class mean %.0f" % (regime, M, name, e0.elapsed_time(e1) / 5 * 1e3, e0.elapsed_time(e1) / 5 * 1e3 / n_cls, nk.float().mean().item()))
lib.mpn_debug_set_nms_force_exact(0)
lib.mpn_debug_set_nms_fused(1)
lib.mpn_debug_set_nms_fused_replay(1)
lib.mpn_debug_set_nms_fused_fence(1)
