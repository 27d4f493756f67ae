// roi_pool.hip — inn.ROIPooling:updateOutput for gfx950 (NCHW module-level form).
//
// HBM-bound: the feature map (C2: 4.9 MB) is L2/MALL resident, the output (N*C*PH*PW fp32, 100 MB at
// C2) is streamed once.  One thread per output element, pw fastest, so a wave writes 256 contiguous
// bytes and reads neighbouring bins of one (roi, channel) plane.  Bin bounds use the same fp32
// operations as the oracle (roundf, floorf, ceilf on fp32 products, no FMA) so argmax is bit-exact.
#include "mpn_internal.h"

namespace mpn {

__global__ __launch_bounds__(256) void roi_pool_nchw_kernel(const float *__restrict__ feat, int B, int C, int H, int W,
                                                            const float *__restrict__ rois, int N, int PH, int PW,
                                                            float scale, RoiRule rr, float *__restrict__ out,
                                                            int32_t *__restrict__ argmax) {
  const size_t total = (size_t)N * C * PH * PW;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    int pw = (int)(t % PW);
    int ph = (int)((t / PW) % PH);
    int c = (int)((t / ((size_t)PW * PH)) % C);
    int n = (int)(t / ((size_t)PW * PH * C));
    const float *ro = rois + 5 * (size_t)n;
    int b = (int)ro[0] - 1;
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    int hs, he, ws, we;
    roi_bin_bounds(ro, scale, rr, H, W, PH, PW, ph, pw, hs, he, ws, we);
    bool empty = (he <= hs) || (we <= ws);
    float m = empty ? 0.0f : -INFINITY;
    int mi = -1;
    const float *fp = feat + ((size_t)b * C + c) * H * W;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        float v = fp[(size_t)h * W + w];
        if (v > m) { m = v; mi = h * W + w; }
      }
    out[t] = m;
    if (argmax) argmax[t] = mi;
  }
}

}  // namespace mpn

using namespace mpn;

extern "C" int mpn_roi_pool_forward_rule(const float *d_feat, int B, int C, int H, int W, const float *d_rois, int N, int PH, int PW,
                                         float scale, float coord_offset, int end_adjust, int bin_rule, float *d_out, int32_t *d_argmax,
                                         void *stream) {
  MPN_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0 && N >= 0);
  MPN_CHECK_ARG(bin_rule == MPN_ROI_BINS_CAFFE || bin_rule == MPN_ROI_BINS_ADAPTIVE);
  if (N == 0) return MPN_OK;
  MPN_CHECK_ARG(d_feat && d_rois && d_out);
  size_t total = (size_t)N * C * PH * PW;
  size_t blocks = cdiv_sz(total, 256);
  if (blocks > 256 * 64) blocks = 256 * 64;  // grid-stride beyond 64 blocks per CU
  hipLaunchKernelGGL(roi_pool_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), d_feat, B, C, H, W, d_rois,
                     N, PH, PW, scale, RoiRule{coord_offset, end_adjust, bin_rule}, d_out, d_argmax);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_roi_pool_forward(const float *d_feat, int B, int C, int H, int W, const float *d_rois, int N, int PH,
                                    int PW, float scale, float coord_offset, int end_adjust, float *d_out,
                                    int32_t *d_argmax, void *stream) {
  return mpn_roi_pool_forward_rule(d_feat, B, C, H, W, d_rois, N, PH, PW, scale, coord_offset, end_adjust, MPN_ROI_BINS_CAFFE, d_out, d_argmax,
                                   stream);
}
