"""timing experiments on the Winograd conv kernel: knock out DMA / barrier / transform / epilogue / fragment loads"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")  # libmpn_hip_dbg.so: the flavour with the mpn_debug_* hooks
lib.mpn_debug_set_conv_variant(7)
for (ci, co, h, w) in [(64, 64, 600, 1000), (128, 128, 300, 500), (512, 512, 38, 63)]:
    for ab in [0, 1, 2, 4, 8, 5, 23, 31]:
        lib.mpn_debug_set_gemm_ablate(ab)
        ms = C.c_float()
        lib.mpn_debug_bench_conv(ci, co, h, w, 0, 10, C.byref(ms))
        print("wino %d->%d %dx%d ablate=%2d (noDMA=%d noBar=%d noTF=%d noEpi=%d noFrag=%d): %.1f us  %.1f TF/s(alg)" % (
            ci, co, h, w, ab, ab & 1, (ab >> 1) & 1, (ab >> 2) & 1, (ab >> 3) & 1, (ab >> 4) & 1, ms.value * 1e3, 2.0 * h * w * ci * 9 * co / ms.value / 1e9))
lib.mpn_debug_set_gemm_ablate(0)
