"""timing experiments on the pixel-major ROI pooling kernel (bench ROIs on a 512 x 38 x 63 map): knock out loads / stores / ROI decode"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multipathnet_amd, bench
lib = multipathnet_amd._lib.load("debug")
_, boxes = bench.synthetic_inputs()
rois = np.concatenate([np.ones((boxes.shape[0], 1), np.float32), boxes], 1).astype(np.float32)
rois = np.ascontiguousarray(rois)
for ab in [0, 1, 2, 3, 4, 6]:
    lib.mpn_debug_set_gemm_ablate(ab)
    ms = C.c_float()
    rc = lib.mpn_debug_bench_roipool(rois.ctypes.data_as(C.c_void_p), rois.shape[0], 512, 38, 63, 20, C.byref(ms))
    print("roi_pool_pm rc=%d ablate=%d (noLoad=%d noStore=%d fixed3x3=%d): %.1f us   %.2f TB/s (100 MB out)" % (
        rc, ab, ab & 1, (ab >> 1) & 1, (ab >> 2) & 1, ms.value * 1e3, 100.4e6 / ms.value / 1e9))
lib.mpn_debug_set_gemm_ablate(0)
