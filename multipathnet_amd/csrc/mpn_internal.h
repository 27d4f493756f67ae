// mpn_internal.h — shared helpers for the libmpn_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/mpn.h"

namespace mpn {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define MPN_CHECK_ARG(cond)                                                     \
  do {                                                                          \
    if (!(cond)) {                                                              \
      ::mpn::set_error("%s: invalid argument: %s", __func__, #cond);            \
      return MPN_EINVAL;                                                        \
    }                                                                           \
  } while (0)

#define MPN_CHECK_HIP(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::mpn::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(_e));          \
      return MPN_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

#define MPN_CHECK_LAUNCH() MPN_CHECK_HIP(hipGetLastError())

constexpr int kWave = 64;

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t cdiv_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// nms.c:14-41 — IoU with the +1 convention, evaluated exactly in the reference's operation order
// (this TU set is built with -ffp-contract=off, so no product is fused into a sum).
__device__ __forceinline__ float iou_plus1(float ax1, float ay1, float ax2, float ay2, float bx1, float by1,
                                           float bx2, float by2) {
  float x1 = ax1 > bx1 ? ax1 : bx1;
  float y1 = ay1 > by1 ? ay1 : by1;
  float x2 = ax2 < bx2 ? ax2 : bx2;
  float y2 = ay2 < by2 ? ay2 : by2;
  float w = x2 - x1 + 1.0f;
  float h = y2 - y1 + 1.0f;
  float inter = w * h;
  float aarea = (ax2 - ax1 + 1.0f) * (ay2 - ay1 + 1.0f);
  float barea = (bx2 - bx1 + 1.0f) * (by2 - by1 + 1.0f);
  float iou = inter / (aarea + barea - inter);
  return (w <= 0.0f || h <= 0.0f) ? 0.0f : iou;
}

}  // namespace mpn
