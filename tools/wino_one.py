"""one Winograd layer (128->128, 300x500) — PMC target for tools/pmc_wino.sh"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
lib.mpn_debug_set_conv_variant(7)
ms = C.c_float()
lib.mpn_debug_bench_conv(128, 128, 300, 500, 0, 10, C.byref(ms))
print("%.1f us" % (ms.value * 1e3))
