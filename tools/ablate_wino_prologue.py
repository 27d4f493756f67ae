"""Round 5, VERDICT r4 task 3(i): before building the persistent Winograd launch with a constant store count, measure its CEILING — the
twelve VGG layers with the prologue's DMA wait knocked out (ablate = 128: what prefetching the next tile's first chunk under the previous
tile's epilogue could hide at the very best; wrong results, timing only) and, for scale, with the epilogue knocked out too (136).
Debug flavour.  python tools/ablate_wino_prologue.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")
lib.mpn_debug_set_conv_variant(7)
LAYERS = [("conv1_2", 64, 64, 600, 1000, 1), ("conv2_1", 64, 128, 300, 500, 0), ("conv2_2", 128, 128, 300, 500, 1), ("conv3_1", 128, 256, 150, 250, 0),
          ("conv3_2", 256, 256, 150, 250, 0), ("conv3_3", 256, 256, 150, 250, 1), ("conv4_1", 256, 512, 75, 125, 0), ("conv4_2", 512, 512, 75, 125, 0),
          ("conv4_3", 512, 512, 75, 125, 1), ("conv5_1", 512, 512, 38, 63, 0), ("conv5_2", 512, 512, 38, 63, 0), ("conv5_3", 512, 512, 38, 63, 0)]
tot = {0: 0.0, 128: 0.0, 136: 0.0}
for name, ci, co, h, w, pool in LAYERS:
    row = []
    for ab in (0, 128, 136):
        lib.mpn_debug_set_gemm_ablate(ab)
        ms = C.c_float()
        lib.mpn_debug_bench_conv(ci, co, h, w, pool, 20, C.byref(ms))
        tot[ab] += ms.value * 1e3
        row.append(ms.value * 1e3)
    print("%-8s %3d->%3d %4dx%-4d: real %7.1f us | prologue DMA not waited for %7.1f us (%+5.1f %%) | + no epilogue %7.1f us" % (
        name, ci, co, h, w, row[0], row[1], 100.0 * (row[1] / row[0] - 1.0), row[2]))
lib.mpn_debug_set_gemm_ablate(0)
print("twelve layers: real %.1f us | prologue DMA wait removed %.1f us (%+.1f %%) | + epilogue removed %.1f us (%+.1f %%)" % (
    tot[0], tot[128], 100.0 * (tot[128] / tot[0] - 1.0), tot[136], 100.0 * (tot[136] / tot[0] - 1.0)))
