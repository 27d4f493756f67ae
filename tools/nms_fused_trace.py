"""s_memtime phase trace of nms_fused_kernel (debug flavour): class 0's last block.  python tools/nms_fused_trace.py [M] [n_cls]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multipathnet_amd
from multipathnet_amd import utils, _lib
from conftest import random_scored_boxes
lib = _lib.load("debug")
lib.mpn_debug_set_nms_fused(2)   # the fused kernel at every size it can take
dev = torch.device("cuda:0")
Ms = [int(sys.argv[1])] if len(sys.argv) > 1 else [300, 1000]
n_cls_list = [int(sys.argv[2])] if len(sys.argv) > 2 else [1, 20]
names = ["keys", "sort", "gather + eq + fw", "mask slice", "arrive", "mask -> LDS", "greedy selection", "write rows"]
for M in Ms:
    for n_cls in n_cls_list:
        for regime in ("distinct", "onetie", "ties", "saturated"):
            rng = np.random.default_rng(0)
            sb = np.stack([random_scored_boxes(rng, M, "distinct" if regime == "onetie" else regime) for _ in range(n_cls)])
            if regime == "onetie":   # the last two boxes class 0 keeps carry the same score: the replay has every earlier round to catch up on
                with _lib.debug_hooks():
                    k0, i0 = utils.nms_with_index(torch.from_numpy(sb[0]).to(dev), 0.3)
                i0 = i0.cpu().numpy()
                sb[0, i0[-1], 4] = sb[0, i0[-2], 4]
            d = torch.from_numpy(sb).to(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with _lib.debug_hooks():
                for _ in range(3):
                    e0.record()
                    keep, idx, nk = utils.nms_batched(d, None, 0.3)
                    e1.record(); torch.cuda.synchronize()
            tr = (C.c_ulonglong * 16)()
            lib.mpn_debug_get_nms_fused_trace(tr)
            t = [int(x) for x in tr]
            tot = t[8] - t[0]
            print("M=%d x %d classes, %-9s: kept %d (ties flag %d), %.1f us by events; class 0's last block %d shader cycles:" % (M, n_cls, regime, t[9], t[10], e0.elapsed_time(e1) * 1e3, tot))
            print("   " + "  ".join("%s %d" % (names[i], t[i + 1] - t[i]) for i in range(8)) + ("   | %.0f cycles / pick" % ((t[7] - t[6]) / max(1, t[9])))
                  + ("   | replay: %d cycles in %d simulate() calls, %d batches" % (t[11], t[12], t[13]) if t[12] else ""))
            S = min(16, max(1, 256 // n_cls), (M + 31) // 32)
            nb = min(4096, S * n_cls)
            w = (C.c_ulonglong * (2 * nb))()
            lib.mpn_debug_get_nms_fused_wall(w, nb)
            w = np.array(list(w), dtype=np.int64).reshape(nb, 2)
            t0 = w[:, 0].min()
            st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
            sel = en.reshape(n_cls, -1).max(axis=1)      # the selecting block of each class ends last
            print("   wall clock (us after the first block's entry), %d blocks: entries min/median/max %.1f/%.1f/%.1f   exits of the non-selecting blocks median %.1f   "
                  "selecting blocks' exits min/median/max %.1f/%.1f/%.1f" % (nb, st.min(), np.median(st), st.max(), np.median(en), sel.min(), np.median(sel), sel.max()))
            if t[12]:
                w8 = (C.c_ulonglong * 8016)()
                lib.mpn_debug_get_nms_fused_wall(w8, 4008)
                g = [int(x) for x in w8[8000:8008]]
                print("      simulate(): death_of %d cycles, round-space preload %d, fixpoint %d, commit %d   (the counters themselves cost ~1 k cycles a batch)" % (g[3], g[4], g[5], g[6]))
            slow = [c for c in range(n_cls) if sel[c] > 1.3 * np.median(sel)]
            for c in slow[:6]:
                sc = np.sort(sb[c, :, 4])
                print("      class %d: exit %.1f us, kept %d, %d bit-equal adjacent score pairs, selecting slice %d" % (c, sel[c], int(nk[c]), int((sc[1:] == sc[:-1]).sum()), int(en.reshape(n_cls, -1)[c].argmax())))
