import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd.load()
for ab in (0, 1, 2, 4, 3, 5, 6, 7):
    lib.mpn_debug_set_gemm_ablate(ab)
    ms = C.c_float()
    lib.mpn_debug_bench_linear(1000, 25088, 4096, 5, C.byref(ms))
    print("ablate=%d (noDMA=%d noBarrier=%d noLDSread=%d): %.1f us  %.1f TF/s" % (ab, ab & 1, (ab >> 1) & 1, (ab >> 2) & 1, ms.value * 1e3, 2.0 * 1000 * 25088 * 4096 / ms.value / 1e9))
