"""one Winograd layer (128->128, 300x500) in block-per-tile (mode 0) and persistent (mode 1) form — PMC target"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd.load()
lib.mpn_debug_set_conv_variant(7)
for mode in (0, 1):
    lib.mpn_debug_set_conv_mode(mode)
    ms = C.c_float()
    lib.mpn_debug_bench_conv(128, 128, 300, 500, 0, 10, C.byref(ms))
    print("mode %d: %.1f us" % (mode, ms.value * 1e3))
