"""s_memtime phase trace of keep_top_k_kernel (debug flavour): 20 classes x ~650 kept rows, k = 100 (the bench image's shape)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n_cls, M = 20, 1000
for name, gen in (("distinct", lambda n: np.sort(rng.random(n).astype(np.float32))[::-1]),
                  ("quantised (64 values)", lambda n: np.sort((rng.integers(1, 65, n) / 64.0).astype(np.float32))[::-1])):
    keep = np.zeros((n_cls, M, 5), np.float32)
    n = rng.integers(550, 750, n_cls).astype(np.int32)
    for c in range(n_cls):
        keep[c, : n[c], 4] = gen(n[c])
    kd, nd = torch.from_numpy(keep).to(dev), torch.from_numpy(n).to(dev)
    out = torch.empty((int(n.sum()), 6), device=dev); thr = torch.zeros(1, device=dev); no = torch.zeros(1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        e0.record()
        rc = lib.mpn_keep_top_k(p(kd), p(nd), n_cls, M, 100, p(thr), p(out), out.shape[0], p(no), None)
        e1.record(); torch.cuda.synchronize()
    tr = (C.c_ulonglong * 8)()
    lib.mpn_debug_get_topk_trace(tr)
    t = [int(x) for x in tr]
    names = ["class offsets", "key staging", "min / max", "radix select", "count", "compaction"]
    tot = t[6] - t[0]
    print("%s: %d keys, kept %d, %.1f us (events; null stream); s_memtime (shader cycles), %d in total:" % (name, int(n.sum()), int(no.item()), e0.elapsed_time(e1) * 1e3, tot))
    for i in range(6):
        print("   %-14s %6d cycles = %4.1f %%" % (names[i], t[i + 1] - t[i], 100.0 * (t[i + 1] - t[i]) / tot))
