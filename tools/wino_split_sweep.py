"""empirical split-K sweep of the Winograd kernel on the VGG layer shapes (calibrates wino_pick_splits)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
lib.mpn_debug_set_conv_variant(7)
shapes = [(64, 64, 600, 1000), (64, 128, 300, 500), (128, 128, 300, 500), (128, 256, 150, 250), (256, 256, 150, 250), (256, 512, 75, 125), (512, 512, 75, 125), (512, 512, 38, 63)]
for (ci, co, h, w) in shapes:
    res = []
    for S in (0, 1, 2, 3, 4, 5, 6, 8):
        if S > ci // 8: continue
        lib.mpn_debug_set_conv_split(S)
        ms = C.c_float()
        lib.mpn_debug_bench_conv(ci, co, h, w, 0, 10, C.byref(ms))
        res.append((S, ms.value * 1e3))
    tiles = ((h + 15) // 16) * ((w + 15) // 16) * ((co + 63) // 64)
    print("%3d->%3d %4dx%-4d tiles %4d chunks %2d: " % (ci, co, h, w, tiles, ci // 8) + "  ".join("S=%d:%.1f" % r for r in res))
lib.mpn_debug_set_conv_split(0)
