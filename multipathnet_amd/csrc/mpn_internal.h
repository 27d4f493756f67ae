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

// ---- library-owned device scratch ---------------------------------------------------------------------------
// Split-K slabs, NMS masks and the like are grown on demand and kept.  They are never process-global: a Scratch
// belongs either to ONE pipeline handle (mpn_frcnn; bound to the calling thread for the duration of an entry point
// by ScratchScope) or, for module-level calls, to ONE (device, stream) pair of a registry that common.hip guards
// with a mutex.  Work on a stream is ordered, so one Scratch per stream of work is race-free; two handles, two host
// threads or two devices never share a buffer (the reference host runs one worker thread per GPU in ONE process,
// test_runner.lua:55-66).
enum ScratchSlot { SCR_CONV_SPLITK = 0, SCR_GEMM_SPLITK, SCR_L2NORM, SCR_NMS, SCR_LINEAR_PACK, SCR_GRAPH_SPLITK, SCR_MISC, SCR_IM2COL, SCR_NMS_CNT, SCR_GEMM_SPLITK_SIDE, SCR_GEMM_SPLITK_LANE, SCR_NSLOTS };
struct Scratch {
  int device = -1;
  void *buf[SCR_NSLOTS] = {};
  size_t bytes[SCR_NSLOTS] = {};
  void release();  // hipFree everything (the owner has synchronised)
};
// The current thread's Scratch for `s`: the ScratchScope override if one is active, else the registry entry of
// (current device, s).  Returns a buffer of at least `need` bytes in *out (grown by sync(s) + free + malloc).
int scratch_get(ScratchSlot slot, size_t need, hipStream_t s, void **out);
// the same, and the buffer is zero-filled (on `s`) whenever it is (re)allocated: self-resetting device counters live here
int scratch_get_zeroed(ScratchSlot slot, size_t need, hipStream_t s, void **out);
struct ScratchScope {
  Scratch *prev;
  explicit ScratchScope(Scratch *sc);
  ~ScratchScope();
};
// Allocation generation: bumped whenever a library-owned buffer that kernels reference is REPLACED (scratch growth, a regrown
// per-image table).  A captured launch graph (pipeline.hip) bakes device pointers into its kernel nodes, so it is only replayed while
// the generation it was captured under is still current.
unsigned long long alloc_generation();
void bump_alloc_generation();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): function attributes are per device.
int set_max_dyn_lds(const void *fn, int bytes);
int project_im_rois_copy(const float *d_boxes, int n, double scale, float *d_rois, float *d_boxes_copy, hipStream_t s);  // boxes.hip
// mpn_nms_batched for a call that runs on a side stream UNDER other work (the pipelined forms' tail under the next image's trunk): keeps the
// launch chain for tables whose fused-kernel blocks (150 KB of LDS each) would displace that work (nms.hip: nms_batched_core)
int nms_batched_under_trunk(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep, int *d_keep_idx,
                            int *d_n_keep, hipStream_t stream);

// ---- tuning knobs ---------------------------------------------------------------------------------------------
// Test / timing hooks (forced kernel variants, split factors, ablations, traces) exist only in the DEBUG flavour of
// the library (make DEBUG_HOOKS=1 -> libmpn_hip_dbg.so, -DMPN_DEBUG_HOOKS).  In the product build every knob is a
// compile-time constant, no mpn_debug_* symbol is exported and the timing-only kernel instantiations are not built.
#ifdef MPN_DEBUG_HOOKS
#define MPN_KNOB(type, name, init) static type name = init
#define MPN_ABLATE(expr) (expr)
#else
#define MPN_KNOB(type, name, init) static constexpr type name = init
#define MPN_ABLATE(expr) 0
#endif

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

// ---- inn.ROIPooling's window and bins, both conventions (include/mpn.h: MPN_ROI_BINS_*; the oracle's orc_roi_pool) -------------------
// bins = 0, the CUDA kernel's rule (Fast R-CNN):  start = round((x1 - off) * scale), end = round((x2 - off) * scale) + end_adjust, window forced
//   >= 1 x 1, bin p = [floor(p * size / P), ceil((p + 1) * size / P)) with the fp32 bin size, offset by the start, THEN clipped to the map
//   (a bin that falls outside is empty -> 0).
// bins = 1, the module's CPU branch (crop + nn.SpatialAdaptiveMaxPooling; alexnet.lua:23 / vgg.lua:28 with float tensors): the window's
//   corners are rounded in the Lua order (x - off) * scale + 1 (1-based map coordinates), CLIPPED to the map FIRST (the reference clips the
//   upper side with cmin and indexes the tensor with the lower side, which must therefore be >= 1: we clip both), and the bins divide the
//   clipped crop: [floor(p * size / P), ceil((p + 1) * size / P)) in integer products — never empty.  The two differ on windows whose
//   rounded corners leave the map (the foveal regions, border boxes) and, once in a thousand, by one cell (fp32 bin size rounding).  All fp32, no contraction (-ffp-contract=off).
struct RoiRule { float coord_offset; int end_adjust; int bins; };
__device__ __forceinline__ void roi_bin_bounds(const float *__restrict__ ro, float scale, RoiRule rr, int H, int W, int PH, int PW, int ph, int pw,
                                               int &hs, int &he, int &ws, int &we) {
  if (rr.bins == 0) {
    const int sw = (int)roundf((ro[1] - rr.coord_offset) * scale);
    const int sh = (int)roundf((ro[2] - rr.coord_offset) * scale);
    const int ew = (int)roundf((ro[3] - rr.coord_offset) * scale) + rr.end_adjust;
    const int eh = (int)roundf((ro[4] - rr.coord_offset) * scale) + rr.end_adjust;
    const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    hs = (int)floorf((float)ph * bh) + sh; he = (int)ceilf((float)(ph + 1) * bh) + sh;
    ws = (int)floorf((float)pw * bw) + sw; we = (int)ceilf((float)(pw + 1) * bw) + sw;
    hs = min(max(hs, 0), H); he = min(max(he, 0), H);
    ws = min(max(ws, 0), W); we = min(max(we, 0), W);
  } else {
    int x1 = (int)roundf((ro[1] - rr.coord_offset) * scale + 1.0f) - 1, y1 = (int)roundf((ro[2] - rr.coord_offset) * scale + 1.0f) - 1;
    int x2 = (int)roundf((ro[3] - rr.coord_offset) * scale + 1.0f) - 1 + rr.end_adjust, y2 = (int)roundf((ro[4] - rr.coord_offset) * scale + 1.0f) - 1 + rr.end_adjust;
    x1 = min(max(x1, 0), W - 1); x2 = min(max(x2, 0), W - 1);
    y1 = min(max(y1, 0), H - 1); y2 = min(max(y2, 0), H - 1);
    const int cw = max(x2 - x1 + 1, 1), ch = max(y2 - y1 + 1, 1);
    hs = (int)floorf((float)(ph * ch) / (float)PH) + y1; he = (int)ceilf((float)((ph + 1) * ch) / (float)PH) + y1;
    ws = (int)floorf((float)(pw * cw) / (float)PW) + x1; we = (int)ceilf((float)((pw + 1) * cw) / (float)PW) + x1;
  }
}

}  // namespace mpn
