import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")
for S in (0, 4, 8, 12, 16, 24, 32, 64):
    lib.mpn_debug_set_gemm_split(S)
    ms = C.c_float()
    rc = lib.mpn_debug_bench_linear(1000, 4096, 105, 50, C.byref(ms))
    print("heads 1000x4096x105 split=%d rc=%d: %.1f us" % (S, rc, ms.value * 1e3))
lib.mpn_debug_set_gemm_split(0)
