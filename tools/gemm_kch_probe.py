"""fc6 / fc7 GEMM with 32-k (kch 4) and 64-k (kch 8) LDS stages"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
for kch in (4, 8):
    lib.mpn_debug_set_gemm_kch(kch)
    for (M, K, N) in [(1000, 25088, 4096), (1000, 4096, 4096)]:
        ms = C.c_float(); lib.mpn_debug_bench_linear(M, K, N, 10, C.byref(ms))
        print("kch=%d M=%d K=%d N=%d %.1f us %.1f TF/s" % (kch, M, K, N, ms.value * 1e3, 2.0 * M * K * N / ms.value / 1e9))
lib.mpn_debug_set_gemm_kch(0)
