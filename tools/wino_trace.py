"""s_memtime trace of the 4-wave Winograd kernel's chunk loop (wave 0 of blocks 0..3): cycles from chunk start to the
pre-barrier point (MFMA pairs 0-6 + everything interleaved), the lgkmcnt(0) wait, and the vmcnt(0)+s_barrier wait.
s_memtime counts shader cycles on gfx950."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
lib.mpn_debug_set_conv_variant(7)
buf = torch.zeros(4 * 64 * 4 + 48 + 4096 * 3, dtype=torch.int64, device="cuda")
lib.mpn_debug_set_wino_trace(C.c_void_p(buf.data_ptr()))
for (ci, co, h, w) in [(64, 64, 600, 1000), (128, 128, 300, 500), (256, 256, 150, 250), (512, 512, 75, 125), (512, 512, 38, 63)]:
    for ab in (64,):
        buf.zero_()
        lib.mpn_debug_set_gemm_ablate(ab)
        ms = C.c_float()
        lib.mpn_debug_bench_conv(ci, co, h, w, 0, 3, C.byref(ms))
        torch.cuda.synchronize()
        pairs = buf.cpu()[4 * 64 * 4:4 * 64 * 4 + 32].view(4, 8)
        phases = buf.cpu()[4 * 64 * 4 + 32:4 * 64 * 4 + 48].view(4, 4)
        blocks = buf.cpu()[4 * 64 * 4 + 48:].view(4096, 3)
        t = buf.cpu()[:4 * 64 * 4].view(4, 64, 4)
        print("wino %d->%d %dx%d: %.1f us/launch" % (ci, co, h, w, ms.value * 1e3))
        for b in range(4):
            rows = [r for r in t[b].tolist() if r[0] != 0]
            if not rows: continue
            t0 = rows[0][0]
            print(" block %d: chunks traced %d; per chunk [start-offset, pairs0-6, lgkm wait, vm+barrier wait] (cycles):" % (b, len(rows)))
            print("   " + " ".join("[%d %d %d %d]" % (r[0] - t0, r[1], r[2], r[3]) for r in rows[:8]))
            pr = pairs[b].tolist()
            print("   block phases [prologue, K loop, epilogue issue]: %s" % phases[b].tolist()[:3])
            print("   chunk 5 pair-start deltas: " + " ".join(str(pr[i + 1] - pr[i]) for i in range(7)))
        # per-CU timeline: gaps between one block's end and the next block's start on the same CU
        import collections
        per_cu = collections.defaultdict(list)
        for r in blocks.tolist():
            if r[1] == 0: continue
            hw, xcc = r[0] & 0xffffffff, r[0] >> 32
            cu = (xcc & 0xf, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
            per_cu[cu].append((r[1], r[2]))
        gaps, durs, counts = [], [], []
        for cu, lst in per_cu.items():
            lst.sort(); counts.append(len(lst))
            durs += [e - b for b, e in lst]
            gaps += [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        if gaps:
            gaps.sort(); durs.sort()
            print("   CUs seen %d, blocks per CU min/max %d/%d; block duration median %d ticks; end->next-start gap median %d, p10 %d, p90 %d" % (
                len(per_cu), min(counts), max(counts), durs[len(durs) // 2], gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[9 * len(gaps) // 10]))
            t_all = [x for lst in per_cu.values() for x in lst]
            print("   kernel span %d ticks" % (max(e for b, e in t_all) - min(b for b, e in t_all)))
lib.mpn_debug_set_gemm_ablate(0)
lib.mpn_debug_set_wino_trace(None)
