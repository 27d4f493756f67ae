import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import multipathnet_amd
lib = multipathnet_amd.load()
for kch in (4, 8):
    for nb in (2, 4):
        lib.mpn_debug_set_gemm_kch(kch); lib.mpn_debug_set_gemm_nbuf(nb)
        for (M, K, N) in [(1000, 25088, 4096), (1000, 4096, 4096)]:
            ms = C.c_float(); lib.mpn_debug_bench_linear(M, K, N, 10, C.byref(ms))
            print("kch=%d nbuf=%d M=%d K=%d N=%d %.1f us %.1f TF/s" % (kch, nb, M, K, N, ms.value * 1e3, 2.0 * M * K * N / ms.value / 1e9))
