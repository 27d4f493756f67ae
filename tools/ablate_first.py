"""timing experiments on the first-layer kernel (conv1_1, K = 36): knock out the output stores / the MFMAs; grid size sweep"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd
lib = multipathnet_amd._lib.load("debug")
def run(ab):
    lib.mpn_debug_set_gemm_ablate(ab)
    ms = C.c_float()
    rc = lib.mpn_debug_bench_conv(3, 64, 600, 1000, 0, 20, C.byref(ms))
    assert rc == 0
    return ms.value * 1e3
for blocks in sys.argv[1:] or ["1280"]:  # 1280 = the product default (-> 1200 blocks x 2 tiles at 600 x 1000)
    os.environ["MPN_FIRST_BLOCKS"] = blocks
    for ab in [0, 1, 2, 3]:
        us = run(ab)
        print("first 3->64 600x1000 blocks<=%s ablate=%d (noStore=%d noMFMA=%d): %.1f us   %.2f TB/s (154 MB out)" % (blocks, ab, ab & 1, (ab >> 1) & 1, us, 153.6e6 / us / 1e6))
lib.mpn_debug_set_gemm_ablate(0)
