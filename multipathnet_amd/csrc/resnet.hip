// resnet.hip — ResNet Fast R-CNN trunk and per-ROI head (models/resnet.lua:24-50) in fp32 on the matrix cores.
//
// One generic implicit-GEMM convolution (any kernel size / stride / pad, batched over maps, fused bias + residual add +
// ReLU) covers conv1 7x7/2, the 1x1 / 3x3 convolutions of the residual blocks, the strided shortcuts and — with the
// batch = the ROIs — the whole per-ROI layer4.  M side = 128 output channels, N side = 128 output pixels of the flattened
// (map, y, x) index space, K step = 8 input channels of one filter tap.  Per step the 4-KiB weight slice is a linear copy of
// the packed weights and the 4-KiB activation slice is GATHERED (one 16-byte load per thread, zero outside the map), both
// staged through registers into double-buffered LDS; every MFMA operand fetch is then the same conflict-free ds_read_b128
// as in dense.hip's kernels.  16 KiB of LDS per block lets several blocks share a CU, which hides the gather latency.
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mpn.h"
#include "resnet.h"

namespace mpn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ActI {  // C8I batch of maps: element (b, c, y, x) at ((c/8) * pitch + (b*H + y)*W + x) * 8 + c%8
  float *p;
  int B, C, H, W;
  const float *planar = nullptr;  // the image only: the same transformed pixels as [3][H][W] planes (the im2col first layer reads these)
  int Cb() const { return (C + 7) / 8; }
  size_t rows() const { return (size_t)B * H * W; }
  size_t pitch() const { return (rows() + 127) / 128 * 128; }  // rows per channel-block plane (a C8 matrix of `rows` rows)
};
static size_t c8i_elems(int B, int C, int H, int W) {  // allocation size: planes rounded up to the GEMM's 128-column tiles
  return (size_t)(round_up(C, 128) / 8) * (((size_t)B * H * W + 127) / 128 * 128) * 8;
}

// ------------------------------------------------------------------------------------------------------------------------
// packed weights: [tap = ky*KW + kx][Cin8/8][CoutP][8], CoutP = round_up(Cout, 128); bias [CoutP]
// ------------------------------------------------------------------------------------------------------------------------
__global__ void pack_conv_generic_kernel(const float *__restrict__ w, const float *__restrict__ b, int Cin, int Cout, int KK, int nch, int CoutP,
                                         float *__restrict__ wpk, float *__restrict__ bpk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < (size_t)CoutP) bpk[t] = (t < (size_t)Cout && b) ? b[t] : 0.0f;
  size_t total = (size_t)KK * nch * CoutP * 8;
  if (t >= total) return;
  const int j = (int)(t & 7);
  size_t r = t >> 3;
  const int co = (int)(r % CoutP); r /= CoutP;
  const int ch = (int)(r % nch);
  const int tap = (int)(r / nch);
  const int ci = ch * 8 + j;
  wpk[t] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f;
}

struct GConvArgs {
  const float *in, *wpk, *bpk, *res;
  float *out;
  int B, Cb_in, H, W, nch;
  size_t pitch_in, pitch_out;  // rows per channel-block plane
  int CoutP, Cb_out, KH, KW, sh, sw, ph, pw, OH, OW, relu;
  long long P;  // B * OH * OW output pixels
  float *part;           // split-K: fp32 partial slabs [split][CoutP/8][pitch_out][8] (conv_splitk_finalize_kernel finishes), else nullptr
  int stages_per_split;
  int norelu_cb0, norelu_cb1;  // channel blocks [cb0, cb1) skip the ReLU (fused sibling convolutions with mixed activations); empty by default
};

// KC = 8-channel chunks per LDS stage (all of the same filter tap): 16 * KC MFMAs per wave between two barriers.  KC = 4
// when the channel count allows it (every layer but the 3-channel stem), KC = 1 otherwise.
template <int KC>
__global__ __launch_bounds__(256) void conv2d_c8i_kernel(GConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][KC * 128 * 8];  // [buffer][A | B][chunk][row][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  // block -> tile: a pixel tile's cout tiles run back to back on ONE XCD (blocks are dealt to the 8 XCDs round-robin): the gathered
  // activations — the big operand of a per-ROI layer (10^5 pixel rows) — enter one L2 once instead of once per cout tile
  const int ny = a.CoutP / 128, nx = (int)((a.P + 127) / 128);
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;
  if (tx >= nx) return;
  const long long p0 = (long long)tx * 128;
  const int cout0 = ty * 128;
  const int OHW = a.OH * a.OW;

  // staging roles: thread = (row tid>>1, 16-byte half tid&1) of both 128 x 8 slices of every chunk of the stage
  const int srow = tid >> 1, sh = tid & 1;
  const long long gpix = p0 + srow;
  const bool gvalid = gpix < a.P;
  const int gb = gvalid ? (int)(gpix / OHW) : 0;
  const int grem = gvalid ? (int)(gpix - (long long)gb * OHW) : 0;
  const int goy = grem / a.OW, gox = grem - goy * a.OW;
  const int iy0 = goy * a.sh - a.ph, ix0 = gox * a.sw - a.pw;
  const size_t plane = a.pitch_in * 8;
  const float *in_b = a.in + (size_t)gb * a.H * a.W * 8 + sh * 4;
  const float *w_t = a.wpk + ((size_t)cout0 + srow) * 8 + sh * 4;
  const size_t w_step = (size_t)a.CoutP * 8;

  const int spt = a.nch / KC;             // stages per tap
  const int nstages = a.KH * a.KW * spt;
  f32x4 ra[KC], rb[KC];
  auto fetch = [&](int st) {  // stage = tap * spt + chunk group
    const int tap = st / spt, cg = st - tap * spt;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int iy = iy0 + ky, ix = ix0 + kx;
    const bool ok = gvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const float *wp = w_t + ((size_t)tap * a.nch + (size_t)cg * KC) * w_step;
    const float *ip = in_b + (size_t)cg * KC * plane + ((size_t)iy * a.W + ix) * 8;
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      ra[q] = *reinterpret_cast<const f32x4 *>(wp + (size_t)q * w_step);
      rb[q] = ok ? *reinterpret_cast<const f32x4 *>(ip + (size_t)q * plane) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      *reinterpret_cast<f32x4 *>(&lds[buf][0][(q * 128 + srow) * 8 + sh * 4]) = ra[q];
      *reinterpret_cast<f32x4 *>(&lds[buf][1][(q * 128 + srow) * 8 + sh * 4]) = rb[q];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  // split-K: blockIdx.z owns a contiguous range of stages (small layers: too few 128 x 128 tiles to fill 256 CUs otherwise)
  const int st0 = a.part ? (int)blockIdx.z * a.stages_per_split : 0;
  const int st1 = a.part ? min(nstages, st0 + a.stages_per_split) : nstages;
  fetch(st0);
  stash(0);
  __syncthreads();
  const int frag = l31 * 8 + half * 4;
  for (int st = st0; st < st1; ++st) {
    const int buf = (st - st0) & 1;
    if (st + 1 < st1) fetch(st + 1);  // global loads in flight under this stage's MFMAs
#pragma unroll
    for (int q = 0; q < KC; ++q) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const f32x4 *>(&lds[buf][0][(q * 128 + wm * 64 + mi * 32) * 8 + frag]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const f32x4 *>(&lds[buf][1][(q * 128 + wn * 64 + ni * 32) * 8 + frag]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
    if (st + 1 < st1) stash(buf ^ 1);
    __syncthreads();
  }

  if (a.part) {  // raw partial sums
    float *slab = a.part + (size_t)blockIdx.z * (a.CoutP / 8) * a.pitch_out * 8;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const long long pix = p0 + wn * 64 + ni * 32 + l31;
      if (pix >= a.P) continue;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e];
          *reinterpret_cast<f32x4 *>(slab + ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4) = v;
        }
    }
    return;
  }
  // epilogue: + bias (+ residual) -> ReLU -> C8I float4 stores
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const long long pix = p0 + wn * 64 + ni * 32 + l31;
    if (pix >= a.P) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
        if (cb >= a.Cb_out) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4);
        const size_t off = ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e] + b4[e];
        if (a.res) v += *reinterpret_cast<const f32x4 *>(a.res + off);
        if (a.relu && !(cb >= a.norelu_cb0 && cb < a.norelu_cb1)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.0f ? 0.0f : v[e];
        }
        *reinterpret_cast<f32x4 *>(a.out + off) = v;
      }
  }
}

// split-K finalize (fp32 graph): sum the slabs in a fixed order (deterministic), + bias (+ residual), ReLU
__global__ void conv_splitk_finalize_kernel(const float *__restrict__ part, int splits, int CbP, int Cb_out, size_t pitch, long long P,
                                            const float *__restrict__ bpk, const float *res, int relu, float *out, int norelu_cb0, int norelu_cb1) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Cb_out * (size_t)P * 2;
  if (t >= total) return;
  const int h = (int)(t & 1);
  const size_t r = t >> 1;
  const size_t pix = r % (size_t)P;
  const int cb = (int)(r / (size_t)P);
  const size_t off = ((size_t)cb * pitch + pix) * 8 + h * 4;
  f32x4 v = *reinterpret_cast<const f32x4 *>(bpk + cb * 8 + h * 4);
  for (int z = 0; z < splits; ++z) v += *reinterpret_cast<const f32x4 *>(part + (size_t)z * CbP * pitch * 8 + off);
  if (res) v += *reinterpret_cast<const f32x4 *>(res + off);
  if (relu && !(cb >= norelu_cb0 && cb < norelu_cb1)) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.0f ? 0.0f : v[e];
  }
  *reinterpret_cast<f32x4 *>(out + off) = v;
}

// nn.SpatialMaxPooling(k,k,s,s,p,p), floor mode, on C8I
__global__ void maxpool2d_c8i_kernel(const float *__restrict__ in, int Cb, int B, int H, int W, size_t pitch_in, int k, int stride, int pad, int OH,
                                     int OW, size_t pitch_out, float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * B * OH * OW * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH); r /= OH;
  const int b = (int)(r % B); const size_t cb = r / B;
  f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) {
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (cb * pitch_in + ((size_t)b * H + iy) * W + ix) * 8 + h * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
    }
  *reinterpret_cast<f32x4 *>(out + (cb * pitch_out + ((size_t)b * OH + oy) * OW + ox) * 8 + h * 4) = m;
}

// image transformer (modules/ImageTransformer.lua:19-33, f64 arithmetic) into a one-map C8I image (channels 3..7 zero)
__global__ void image_transform_c8i_kernel(const float *__restrict__ in, int H, int W, int s0, int s1, int s2, double scale, double m0,
                                           double m1, double m2, double d0, double d1, double d2, int has_std, float *__restrict__ out,
                                           float *__restrict__ planar) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t plane = (size_t)H * W;
  if (t >= plane) return;
  double v0 = (double)in[(size_t)s0 * plane + t], v1 = (double)in[(size_t)s1 * plane + t], v2 = (double)in[(size_t)s2 * plane + t];
  if (scale != 1.0) { v0 = v0 * scale; v1 = v1 * scale; v2 = v2 * scale; }
  v0 = v0 + (-m0); v1 = v1 + (-m1); v2 = v2 + (-m2);
  if (has_std) { v0 = v0 / d0; v1 = v1 / d1; v2 = v2 / d2; }
  *reinterpret_cast<f32x4 *>(out + t * 8) = f32x4{(float)v0, (float)v1, (float)v2, 0.0f};
  *reinterpret_cast<f32x4 *>(out + t * 8 + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if (planar) { planar[t] = (float)v0; planar[plane + t] = (float)v1; planar[2 * plane + t] = (float)v2; }
}

// inn.ROIPooling on a one-map C8I feature -> [N][Cb][PH][PW][8] (the batch the per-ROI head convolves); the bin arithmetic
// is the same as roi_pool_c8_kernel / the oracle's orc_roi_pool (coord_offset 1, end_adjust 0)
__global__ void roi_pool_c8i_kernel(const float *__restrict__ feat, int Cb, int H, int W, size_t pitch_f, const float *__restrict__ rois, int roi_stride,
                                    int N, int PH, int PW, float scale, float *__restrict__ out, size_t pitch_o, int roi_bins) {
  const int PP = PH * PW;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * Cb * PP * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int bin = (int)(r % PP); r /= PP;
  const int cb = (int)(r % Cb); const int n = (int)(r / Cb);
  const int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, RoiRule{1.0f, 0, roi_bins}, H, W, PH, PW, ph, pw, hs, he, ws, we);
  const bool empty = (he <= hs) || (we <= ws);
  f32x4 m = empty ? f32x4{0, 0, 0, 0} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const float *fp = feat + (size_t)cb * pitch_f * 8 + h * 4;
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(fp + ((size_t)y * W + x) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
    }
  *reinterpret_cast<f32x4 *>(out + ((size_t)cb * pitch_o + ((size_t)n * PH + ph) * PW + pw) * 8 + h * 4) = m;
}

// the same pooling with the bin arithmetic done once per (roi, bin): a thread owns one output row for CBG channel blocks, reads
// whole 32-byte records and a wave stores 64 consecutive records (the kernel above spends most of its instructions on the
// per-thread bin arithmetic for 16 output bytes).  Same cells, same comparisons: bit-identical.
template <int CBG>
__global__ __launch_bounds__(256) void roi_pool_c8i_rows_kernel(const float *__restrict__ feat, int H, int W, size_t pitch_f, const float *__restrict__ rois,
                                                                 int roi_stride, int N, int PH, int PW, float scale, float *__restrict__ out, size_t pitch_o,
                                                                 int fc_mp, int roi_bins) {
  const int PP = PH * PW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * PP) return;
  const int n = (int)(t / PP), bin = (int)(t - (size_t)n * PP);
  const int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, RoiRule{1.0f, 0, roi_bins}, H, W, PH, PW, ph, pw, hs, he, ws, we);
  const bool empty = (he <= hs) || (we <= ws);
  const int cb0 = blockIdx.y * CBG;
  const float *fp = feat + (size_t)cb0 * pitch_f * 8;
  // fc_mp > 0: the (bin, roi)-row C8 matrix [Cb][PP][fc_mp][8] a fully-connected first layer reads as its GEMM operand
  // (plane = (channel block, bin), row = roi — the VGG pipeline's fc6 layout), instead of the C8I batch of maps
  float *op = fc_mp > 0 ? out + (((size_t)cb0 * PP + bin) * fc_mp + n) * 8 : out + ((size_t)cb0 * pitch_o + t) * 8;
  const size_t plane_o = fc_mp > 0 ? (size_t)PP * fc_mp : pitch_o;
  const float lowest = empty ? 0.0f : -INFINITY;
  f32x4 lo[CBG], hi[CBG];
#pragma unroll
  for (int c = 0; c < CBG; ++c) lo[c] = hi[c] = f32x4{lowest, lowest, lowest, lowest};
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {
      const float *px = fp + ((size_t)y * W + x) * 8;
#pragma unroll
      for (int c = 0; c < CBG; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(px + (size_t)c * pitch_f * 8), b = *reinterpret_cast<const f32x4 *>(px + (size_t)c * pitch_f * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[c][e] = a[e] > lo[c][e] ? a[e] : lo[c][e]; hi[c][e] = b[e] > hi[c][e] ? b[e] : hi[c][e]; }
      }
    }
#pragma unroll
  for (int c = 0; c < CBG; ++c) {
    *reinterpret_cast<f32x4 *>(op + (size_t)c * plane_o * 8) = lo[c];
    *reinterpret_cast<f32x4 *>(op + (size_t)c * plane_o * 8 + 4) = hi[c];
  }
}

// 7x7 global average pool of [N][Cb][H][W][8] into the C8 matrix [Cb][Mp][8] the head GEMM reads (row = roi); the sum
// runs in row-major order like the oracle's, then * 1/(H*W)
__global__ void avgpool_c8i_to_c8_kernel(const float *__restrict__ in, int N, int Cb, int HW, size_t pitch, float inv, float *__restrict__ out,
                                         int Mp) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * Cb * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int n = (int)(r % N); const int cb = (int)(r / N);
  const float *ip = in + ((size_t)cb * pitch + (size_t)n * HW) * 8 + h * 4;
  f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < HW; ++i) sacc += *reinterpret_cast<const f32x4 *>(ip + (size_t)i * 8);
  *reinterpret_cast<f32x4 *>(out + ((size_t)cb * Mp + n) * 8 + h * 4) = sacc * inv;
}

// ------------------------------------------------------------------------------------------------------------------------
// bf16 variant (SURVEY §8f rank 3 asks for the ResNet / Inception trunks in bf16): activations and weights stored as bf16
// (round-to-nearest-even), products accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (K = 16 per instruction = two 8-channel
// records: lanes 0-31 fetch the record of chunk 2q, lanes 32-63 that of chunk 2q+1 — the same split for both operands, so
// the k order inside the instruction cannot matter), bias / residual / ReLU in fp32, one rounding to bf16 at the store.
// Layout: C8I with 16-byte records [C/8][rows, pitch][8 x bf16]; channel-block count padded to even (zero planes).
// ------------------------------------------------------------------------------------------------------------------------
typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// fp32 -> bf16, round to nearest even: gfx950's own conversion (v_cvt_pk_bf16_f32, two values per instruction).  Every kernel of this
// file rounds through these two helpers, so kernels that share a K order stay bit-identical to each other.  (The bit-arithmetic form
// — add 0x7fff + lsb, shift — costs 4-5 VALU instructions per value; at 128 accumulators per lane that was 10-20 % of a convolution
// launch: VALU work never overlaps the wave's MFMAs on this chip.)
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float cvtf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {  // bits 0-15 = bf16(lo), bits 16-31 = bf16(hi)
  const cvtf32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hwbf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }
// ReLU switched by a wave-uniform flag, one v_med3_f32: clamp(v, lo, +inf) with lo = 0 (ReLU) or -inf (none)
__device__ __forceinline__ float relu_lo(bool on) { return on ? 0.0f : -__builtin_inff(); }
__device__ __forceinline__ float clamp_lo(float v, float lo) { return __builtin_amdgcn_fmed3f(v, lo, __builtin_inff()); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }

// packed bf16 weights: [tap][nch2][CoutP][8], nch2 = Cin chunks rounded up to even
__global__ void pack_conv_bf16_kernel(const float *__restrict__ w, int Cin, int Cout, int KK, int nch2, int CoutP, bf16_t *__restrict__ wpk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)KK * nch2 * CoutP * 8;
  if (t >= total) return;
  const int j = (int)(t & 7);
  size_t r = t >> 3;
  const int co = (int)(r % CoutP); r /= CoutP;
  const int ch = (int)(r % nch2);
  const int tap = (int)(r / nch2);
  const int ci = ch * 8 + j;
  wpk[t] = f2bf((co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * KK + tap] : 0.0f);
}

struct GConvArgsB {
  const bf16_t *in, *wpk, *res;
  const float *bpk;
  bf16_t *out;
  int B, H, W, nch2;
  size_t pitch_in, pitch_out;
  int CoutP, Cb_out, KH, KW, sh, sw, ph, pw, OH, OW, relu;
  long long P;
  float *part;           // split-K (128 x 128 kernel only): fp32 partial slabs [split][CoutP/8][pitch_out][8], else nullptr
  int stages_per_split;
  unsigned long long *trace;  // tools/dma_trace.py: s_memtime stamps of wave 0 of block 0 (LDS-DMA kernel), else nullptr
  int norelu_cb0, norelu_cb1;  // channel blocks [cb0, cb1) skip the ReLU (fused sibling convolutions with mixed activations); empty by default
  int exp;                     // LDS-DMA kernel: scheduling experiments (mpn_debug_set_bf16_exp; 0 in the product): bit 0 = s_setprio 1 over the MFMA clusters
};

// KP = chunk PAIRS (16 input channels) per LDS stage: 4 * KP MFMAs per wave between two barriers (instantiated: 1 and 2)
template <int KP>
__global__ __launch_bounds__(256) void conv2d_c8i_bf16_kernel(GConvArgsB a) {
  constexpr int NCH = 2 * KP;                       // 8-channel chunks per stage
  constexpr int NREC = NCH * 128 / 256;             // 16-byte records per thread per operand per stage
  __shared__ __attribute__((aligned(16))) u32x4 lds[2][2][NCH * 128];  // [buffer][A | B][chunk][row] records
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  // block -> tile: a pixel tile's cout tiles run back to back on ONE XCD (blocks are dealt to the 8 XCDs round-robin): the gathered
  // activations — the big operand of a per-ROI layer (10^5 pixel rows) — enter one L2 once instead of once per cout tile
  const int ny = a.CoutP / 128, nx = (int)((a.P + 127) / 128);
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;
  if (tx >= nx) return;
  const long long p0 = (long long)tx * 128;
  const int cout0 = ty * 128;
  const int OHW = a.OH * a.OW;

  // staging roles: thread = row tid & 127, chunks (tid >> 7) + 2 i
  const int srow = tid & 127, sc0 = tid >> 7;
  const long long gpix = p0 + srow;
  const bool gvalid = gpix < a.P;
  const int gb = gvalid ? (int)(gpix / OHW) : 0;
  const int grem = gvalid ? (int)(gpix - (long long)gb * OHW) : 0;
  const int goy = grem / a.OW, gox = grem - goy * a.OW;
  const int iy0 = goy * a.sh - a.ph, ix0 = gox * a.sw - a.pw;
  const u32x4 *in_r = reinterpret_cast<const u32x4 *>(a.in) + (size_t)gb * a.H * a.W;   // record units
  const u32x4 *w_r = reinterpret_cast<const u32x4 *>(a.wpk) + (size_t)cout0 + srow;

  const int spt = a.nch2 / NCH;
  const int nstages = a.KH * a.KW * spt;
  u32x4 ra[NREC], rb[NREC];
  auto fetch = [&](int st) {
    const int tap = st / spt, cg = st - tap * spt;
    const int ky = tap / a.KW, kx = tap - ky * a.KW;
    const int iy = iy0 + ky, ix = ix0 + kx;
    const bool ok = gvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      const int ch = cg * NCH + sc0 + 2 * i;
      ra[i] = w_r[((size_t)tap * a.nch2 + ch) * a.CoutP];
      rb[i] = ok ? in_r[(size_t)ch * a.pitch_in + (size_t)iy * a.W + ix] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NREC; ++i) {
      lds[buf][0][(sc0 + 2 * i) * 128 + srow] = ra[i];
      lds[buf][1][(sc0 + 2 * i) * 128 + srow] = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  // split-K: blockIdx.z owns a contiguous range of stages (small layers: too few 128 x 128 tiles to fill 256 CUs otherwise)
  const int st0 = a.part ? (int)blockIdx.z * a.stages_per_split : 0;
  const int st1 = a.part ? min(nstages, st0 + a.stages_per_split) : nstages;
  fetch(st0);
  stash(0);
  __syncthreads();
  for (int st = st0; st < st1; ++st) {
    const int buf = (st - st0) & 1;
    if (st + 1 < st1) fetch(st + 1);
#pragma unroll
    for (int q = 0; q < KP; ++q) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const bf16x8 *>(&lds[buf][0][(2 * q + half) * 128 + wm * 64 + mi * 32 + l31]);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const bf16x8 *>(&lds[buf][1][(2 * q + half) * 128 + wn * 64 + ni * 32 + l31]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    if (st + 1 < st1) stash(buf ^ 1);
    __syncthreads();
  }

  if (a.part) {  // raw fp32 partial sums; conv_splitk_finalize_bf16_kernel adds bias / residual, applies ReLU and rounds
    float *slab = a.part + (size_t)blockIdx.z * (a.CoutP / 8) * a.pitch_out * 8;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const long long pix = p0 + wn * 64 + ni * 32 + l31;
      if (pix >= a.P) continue;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e];
          *reinterpret_cast<f32x4 *>(slab + ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4) = v;
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const long long pix = p0 + wn * 64 + ni * 32 + l31;
    if (pix >= a.P) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
        if (cb >= a.Cb_out) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4);
        const size_t off = ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4;  // bf16 elements
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e] + b4[e];
        if (a.res) {
          const u16x4 r4 = *reinterpret_cast<const u16x4 *>(a.res + off);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
        }
        u16x4 o;
        const bool rl = a.relu && !(cb >= a.norelu_cb0 && cb < a.norelu_cb1);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf((rl && v[e] < 0.0f) ? 0.0f : v[e]);
        *reinterpret_cast<u16x4 *>(a.out + off) = o;
      }
  }
}

// split-K finalize: sum the slabs in a fixed order (deterministic), + bias (+ residual), ReLU, round to bf16
__global__ void conv_splitk_finalize_bf16_kernel(const float *__restrict__ part, int splits, int CbP, int Cb_out, size_t pitch, long long P,
                                                 const float *__restrict__ bpk, const bf16_t *res, int relu, bf16_t *out, int norelu_cb0, int norelu_cb1) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Cb_out * (size_t)P * 2;
  if (t >= total) return;
  const int h = (int)(t & 1);
  const size_t r = t >> 1;
  const size_t pix = r % (size_t)P;
  const int cb = (int)(r / (size_t)P);
  const size_t off = ((size_t)cb * pitch + pix) * 8 + h * 4;
  f32x4 v = *reinterpret_cast<const f32x4 *>(bpk + cb * 8 + h * 4);
  for (int z = 0; z < splits; ++z) {
    const f32x4 p4 = *reinterpret_cast<const f32x4 *>(part + (size_t)z * CbP * pitch * 8 + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += p4[e];
  }
  if (res) {
    const u16x4 r4 = *reinterpret_cast<const u16x4 *>(res + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bf2f(r4[e]);
  }
  u16x4 o;
  const bool rl = relu && !(cb >= norelu_cb0 && cb < norelu_cb1);
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf((rl && v[e] < 0.0f) ? 0.0f : v[e]);
  *reinterpret_cast<u16x4 *>(out + off) = o;
}

// ---- large layers: 256 couts x 256 pixels per block, operands gathered straight into LDS by the DMA engine ----------------
// The 128 x 128 kernel above loads 1 operand byte per 64 FLOP and is bound by that (L2 -> CU), and its register-staged
// prefetch is one stage deep.  Here a wave owns 128 x 128 (4 x 4 MFMA tiles, 256 accumulator registers): 128 FLOP per operand
// byte; a stage is 32 input channels of one tap (16 KiB of weights + 16 KiB of gathered pixels) and lives in a 4-deep LDS
// ring filled by global_load_lds THREE stages ahead (~3000 cycles of cover with one block per CU) — no staging registers, no
// ds_write.  Lane = one row of the stage: wave w DMAs rows 64 w .. 64 w + 63 of each of the 4 weight chunks (uniform base +
// lane * 16) and of the 4 pixel chunks (uniform chunk-plane base + the lane's gathered pixel offset).  A lane whose tap falls
// outside its map still issues its loads (from its map's first pixel: the per-wave vmcnt bookkeeping needs a fixed number of
// loads per stage) and overwrites its four records with zeros once they have landed, before the barrier that publishes the stage.
__device__ __forceinline__ void glds16_s(const void *base_uniform, unsigned lane_byte_off, unsigned lds_byte_addr_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds_byte_addr_uniform), "v"(lane_byte_off), "s"(base_uniform) : "memory");
}

// MI x NI = 32 x 32 MFMA tiles per wave along the cout / pixel axes; the block is 2 x 2 waves = 64 MI couts x 64 NI pixels.
//   <4, 4>: 256 x 256, 4-deep ring (128 KiB), all 512 registers, one block per CU;
//   <4, 2>: 256 couts x 128 pixels and <2, 4>: 128 couts x 256 pixels: 3-deep ring (72 KiB), <= 256 registers, TWO blocks per CU —
//           one block's prologue / epilogue (~10 us with nothing to overlap them at one block per CU) runs under the other's K
//           loop.  The host picks per layer (rn_conv): cout counts that are not multiples of 256 (Inception's 320 / 384) take
//           <2, 4>, and the narrow shapes also pack some grids into fuller rounds (a 512-cout layer on 49 000 pixels is 384
//           blocks = 1.5 rounds of 256 at 256 x 256).
//   NCH = 8-channel chunks per stage: 4 (32 channels; the shapes above) or 8 (64 channels — half the barriers / waits / zero-fills per
//           MFMA cycle; the stage doubles, so ONE block per CU: <4, 4, 8> two 64-KiB buffers, <4, 2, 8> / <2, 4, 8> three 48-KiB ones).
//   WN = waves along the pixel axis (2 along the cout axis always).  WN = 4 (round 4, <4, 2, 4, 4>, debug flavour): EIGHT waves share a 256 x 256 tile, each
//           128 couts x 64 pixels — the bytes per FLOP of <4, 4> (7.6 KiB of operands per MFLOP through the vector-memory path instead of 11.4) with two waves
//           per SIMD and 128 accumulator registers instead of one wave with 256; 4-deep ring (128 KiB), one block per CU, four DMA items per wave per stage,
//           dealt one every second MFMA.  Bit-identical; measured no faster than the per-layer pick on configs[3]'s layers (1.5 block rounds).
template <int MI, int NI, int NCH = 4, int WN = 2>
__global__ __launch_bounds__(128 * WN, ((MI == 4 && NI == 4) || NCH == 8 || WN == 4) ? 1 : 2) void conv2d_c8i_bf16_dma_kernel(GConvArgsB a, int nx, int ny) {
  static_assert((MI == 4 || MI == 2) && (NI == 4 || NI == 2) && MI + NI >= 6 && (NCH == 4 || NCH == 8) && (WN == 2 || (WN == 4 && MI == 4 && NI == 2 && NCH == 4)), "");
  constexpr int NT = 128 * WN;                                 // threads
  constexpr int TM = 64 * MI, TN = 32 * NI * WN, NQ = NCH / 2;  // NQ = k-steps (16 channels) per stage
  constexpr bool BIG = (MI == 4 && NI == 4) || WN == 4;         // 256 x 256 tiles
  constexpr int RING = NCH == 8 ? (BIG ? 2 : 3) : (BIG ? 4 : 3), LOOK = RING - 1;  // stages in the ring / stages the DMA runs ahead
  constexpr int NA = NCH * TM / NT, NB = NCH * TN / NT;  // weight- / pixel-chunk DMA items per wave per stage (a wave-load = 64 rows of one chunk)
  constexpr int ITEMS = NA + NB;
  constexpr unsigned OPA = NCH * TM * 16, OPBB = NCH * TN * 16, STAGEB = OPA + OPBB;  // bytes: weights / pixels / stage
  extern __shared__ __attribute__((aligned(16))) u32x4 ring[];  // [RING][A: NCH x TM rows | B: NCH x TN rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  // block -> tile: a pixel tile's ny cout tiles run back to back on ONE XCD (blocks are dealt to the 8 XCDs round-robin), so the
  // gathered pixels are fetched into one L2 once
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;
  if (tx >= nx) return;
  const long long p0 = (long long)tx * TN;
  const int cout0 = ty * TM;
  const int OHW = a.OH * a.OW;
  // DMA roles (one wave-load = 64 rows of one 8-channel chunk).  A 256-row operand: wave w moves rows 64 w .. 64 w + 63 of all 4
  // chunks; a 128-row operand: wave w moves rows 64 (w & 1) .. of chunks 2 (w >> 1) and 2 (w >> 1) + 1.
  constexpr int WA = TM / 64, WB = TN / 64;                      // waves per chunk of either operand (a wave-load = 64 rows of one chunk)
  const int arow = (wave % WA) * 64 + lane;                      // this lane's weight row (cout) of the tile
  const int ach0 = NA * (wave / WA);                             // its first weight chunk (wave-uniform)
  const int prow = (wave % WB) * 64 + lane;                      // this lane's pixel row of the tile
  const int bch0 = NB * (wave / WB);                             // its first pixel chunk (wave-uniform)
  const long long gpix = p0 + prow;
  const bool gvalid = gpix < a.P;
  const int gb = gvalid ? (int)(gpix / OHW) : 0;
  const int grem = gvalid ? (int)(gpix - (long long)gb * OHW) : 0;
  const int goy = grem / a.OW, gox = grem - goy * a.OW;
  const int iy0 = goy * a.sh - a.ph, ix0 = gox * a.sw - a.pw;
  const unsigned map_off = (unsigned)gb * (unsigned)(a.H * a.W);  // records
  const unsigned ring0 = (unsigned)(size_t)((__attribute__((address_space(3))) const u32x4 *)ring);
  const unsigned lds_a = ring0 + (unsigned)ach0 * (TM * 16u) + (unsigned)(wave % WA) * 1024u;
  const unsigned lds_b = ring0 + OPA + (unsigned)bch0 * (TN * 16u) + (unsigned)(wave % WB) * 1024u;
  const unsigned a_lane = (unsigned)arow * 16u;
  const int spt = a.nch2 / NCH;
  const int nstages = a.KH * a.KW * spt;

  // issue cursor (the stage being DMA'd): tap (ky, kx), channel group cg and two running chunk pointers — the packed weights are
  // [tap][chunk][CoutP][8], i.e. consecutive stages are consecutive memory; the pixel chunk planes restart at every tap
  int i_cg = 0, i_kx = 0, i_ky = 0;
  const size_t w_step = (size_t)a.CoutP * 16, b_step = a.pitch_in * 16;
  const char *i_wp = reinterpret_cast<const char *>(a.wpk + (size_t)cout0 * 8) + (size_t)ach0 * w_step;
  const char *const b_base = reinterpret_cast<const char *>(a.in) + (size_t)bch0 * b_step;
  const char *i_bp = b_base;
  unsigned i_blane = 0, i_slot = 0;
  auto issue_begin = [&](int slot) -> bool {  // per-lane part (the only VALU work of a stage)
    const int iy = iy0 + i_ky, ix = ix0 + i_kx;
    const bool ok = gvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    i_blane = ok ? (map_off + (unsigned)(iy * a.W + ix)) * 16u : map_off * 16u;
    i_slot = (unsigned)slot * STAGEB;
    return ok;
  };
#ifdef MPN_BF16_ABLATE  // timing experiments (make FLAGS+=-DMPN_BF16_ABLATE; tools/bench_conv_bf16.py; garbage results with bits 1-6; the branches cost the K loop ~13 %, so they are in NO default build): bit 0 = s_setprio 1 over the MFMA clusters,
  const int ex = a.exp;  // bit 1 = no pixel-gather DMA, bit 2 = no weight DMA, bit 3 = no MFMAs, bit 4 = no epilogue loads / stores,
                         // bit 5 = no vmcnt wait / zero-fill / barrier in the K loop, bit 6 = no fragment reads in the K loop
#else
  constexpr int ex = 0;
#endif
  auto issue_a = [&](int k) { if (!(ex & 4)) glds16_s(i_wp, a_lane, lds_a + i_slot + (unsigned)k * (TM * 16u)); i_wp += w_step; };
  auto issue_b = [&](int k) { if (!(ex & 2)) glds16_s(i_bp, i_blane, lds_b + i_slot + (unsigned)k * (TN * 16u)); i_bp += b_step; };
  auto issue_item = [&](int i) {  // weight and pixel chunks alternate while both last
    constexpr int M = NA < NB ? NA : NB;
    if (i < 2 * M) { if ((i & 1) == 0) issue_a(i >> 1); else issue_b(i >> 1); }
    else if (NA > NB) issue_a(i - M);
    else issue_b(i - M);
  };
  auto issue_end = [&]() {
    i_wp += (NCH - NA) * w_step;
    i_bp += (NCH - NB) * b_step;
    if (++i_cg == spt) { i_cg = 0; i_bp = b_base; if (++i_kx == a.KW) { i_kx = 0; ++i_ky; } }
  };
  auto issue_all = [&](int slot) -> bool {
    const bool ok = issue_begin(slot);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) issue_item(i);
    issue_end();
    return ok;
  };
  auto zero_oob = [&](int slot, bool ok) {  // this lane's own NB records of the stage
    if (!ok) {
      u32x4 *B = ring + (size_t)slot * (STAGEB / 16) + OPA / 16 + bch0 * TN + prow;
#pragma unroll
      for (int i = 0; i < NB; ++i) B[i * TN] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto wait_landed = [&](auto in_flight_tag) {  // all but the newest `in flight` stages' loads of this wave
    constexpr int n = decltype(in_flight_tag)::value * ITEMS;
    static_assert(n == 0 || n == 4 || n == 6 || n == 8 || n == 12 || n == 16 || n == 24, "");
    if constexpr (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (n == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  auto next_slot = [](int sl) { return sl + 1 == RING ? 0 : sl + 1; };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  // operand fragments, double-buffered one k-step (16 channels) ahead
  bf16x8 af[2][MI], bf[2][NI];
  const int frag_row_a = half * TM + wm * (MI * 32) + l31, frag_row_b = OPA / 16 + half * TN + wn * (NI * 32) + l31;
  auto load_frags = [&](int slot, int q, int fs) {
    if (ex & 64) return;
    const u32x4 *S = ring + (size_t)slot * (STAGEB / 16);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[fs][mi] = *reinterpret_cast<const bf16x8 *>(S + q * 2 * TM + frag_row_a + mi * 32);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[fs][ni] = *reinterpret_cast<const bf16x8 *>(S + q * 2 * TN + frag_row_b + ni * 32);
  };

  // okn[k]: in-bounds flag of stage st + 1 + k (issued, not yet published); slots: stage st lives in ring slot st % RING
  bool okn[3] = {false, false, false};
  static_assert(LOOK >= 1 && LOOK <= 3, "");
  const bool ok0 = issue_all(0);  // the prologue issues stages 0 .. LOOK - 1
  if (LOOK > 1 && nstages > 1) okn[0] = issue_all(1);
  if (LOOK > 2 && nstages > 2) okn[1] = issue_all(2);
  if constexpr (LOOK > 2) {
    if (nstages > 2) wait_landed(I2{});
    else if (nstages > 1) wait_landed(I1{});
    else wait_landed(I0{});
  } else if constexpr (LOOK > 1) {
    if (nstages > 1) wait_landed(I1{});
    else wait_landed(I0{});
  } else {
    wait_landed(I0{});
  }
  zero_oob(0, ok0);
  __syncthreads();
  load_frags(0, 0, 0);
  int s_cur = 0;                              // slot of stage st
  int s_iss = nstages > LOOK ? LOOK : 0;      // slot of stage st + LOOK

#ifdef MPN_DEBUG_HOOKS
  unsigned long long *const tr = (a.trace && blockIdx.x == 0 && tid == 0) ? a.trace + 8 : nullptr;  // [stage][q0 start, before wait, after wait, after barrier]
#else
  constexpr unsigned long long *tr = nullptr;  // the product kernel carries no trace branches in its K loop
#endif
  int tr_n = 0;
  if (tr) a.trace[0] = __builtin_amdgcn_s_memtime();
  // One stage = 2 k-steps x MI NI MFMAs, hand-scheduled like gemm_c8_pf_kernel (dense.hip): VALU work from a wave does not overlap
  // its own MFMAs but LDS reads, SALU and DMA issue do, so the next k-step's fragment reads follow the FIRST MFMA of a k-step, the
  // DMA items of stage st+LOOK follow the next MFMAs of k-step 0, and the stage barrier (wait for this wave's stage-st+1 loads,
  // zero its out-of-map records, s_barrier) sits BEFORE k-step 1's MFMAs — whose operands are already in registers — so that the
  // next stage's first fragments are fetched under them.
  auto body = [&](auto issue_tag, auto in_flight_tag, auto next_tag) {
    constexpr bool ISSUE = decltype(issue_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value;
    const int s_nxt = next_slot(s_cur);
    const bool prio = (ex & 1) != 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q == NQ - 1 && NEXT) {
        if (tr && tr_n < 96) tr[tr_n * 4 + 1] = __builtin_amdgcn_s_memtime();
        if (!(ex & 32)) wait_landed(in_flight_tag);
        if (tr && tr_n < 96) tr[tr_n * 4 + 2] = __builtin_amdgcn_s_memtime();
        if (!(ex & 32)) zero_oob(s_nxt, okn[0]);
        if (!(ex & 32)) __syncthreads();
        if (tr && tr_n < 96) tr[tr_n * 4 + 3] = __builtin_amdgcn_s_memtime();
        load_frags(s_nxt, 0, 0);
      }
      if (q == 0 && tr && tr_n < 96) tr[tr_n * 4] = __builtin_amdgcn_s_memtime();
      if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < MI * NI; ++t) {
        const int mi = t / NI, ni = t % NI;
        const int gt = q * (MI * NI) + t;  // MFMA index within the stage: the DMA items of stage st + LOOK follow MFMAs 1 .. ITEMS
        __builtin_amdgcn_sched_barrier(0);
        if (!(ex & 8)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1][mi], bf[q & 1][ni], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < NQ && t == 0) load_frags(s_cur, q + 1, (q + 1) & 1);
        if constexpr (ISSUE) {
          static_assert(ITEMS + 1 <= (NQ - 1) * MI * NI, "the DMA issue must end before the stage barrier");
          if (gt == 1) okn[LOOK - 1] = issue_begin(s_iss);
          if constexpr (WN == 4) {  // dealt: one DMA item every second MFMA of k-step 0 (tools/probes/vmem_path_bw.cpp: clumps cost the matrix pipe)
            if ((gt & 1) == 1 && (gt >> 1) < ITEMS) issue_item(gt >> 1);
            if (gt == 2 * ITEMS - 1) issue_end();
          } else {
            if (gt >= 1 && gt <= ITEMS) issue_item(gt - 1);
            if (gt == ITEMS) issue_end();
          }
        }
      }
      if (prio) __builtin_amdgcn_s_setprio(0);
    }
    okn[0] = okn[1]; okn[1] = okn[2];
    s_cur = s_nxt; s_iss = next_slot(s_iss);
    ++tr_n;
  };
  using T = std::true_type;
  using F = std::false_type;
  {
    int st = 0;
    for (; st + LOOK < nstages; ++st) body(T{}, std::integral_constant<int, LOOK - 1>{}, T{});  // the newer stages may still be in flight
    if constexpr (LOOK == 3) {
      if (st + 2 < nstages) { body(F{}, I1{}, T{}); ++st; }
    }
    if (st + 1 < nstages) { body(F{}, I0{}, T{}); ++st; }
    body(F{}, I0{}, F{});
  }

  // Epilogue.  (1) On this ISA stores count in vmcnt too, so a load issued after a store waits for that store's acknowledgement:
  // interleaved bias / residual loads and stores are exposed memory round trips (64 of them took ~50 us per block, more than the
  // whole K loop of a 1x1 layer).  The loads of channel group mi + 1 are therefore issued BEFORE the stores of group mi.  (2) The
  // MFMA layout leaves a pixel's 8-channel record split across lanes l and l + 32 (4 channels = 8 bytes each); the store tail is
  // bound by store INSTRUCTIONS, not bytes, so pairs of channel blocks are exchanged with v_permlane32_swap (lanes 0-31 end up
  // with the whole record of block 2p, lanes 32-63 with block 2p + 1) and written / read as 16-byte accesses.
  if (tr) a.trace[1] = __builtin_amdgcn_s_memtime();
  if (ex & 16) {  // keep the accumulators alive without the epilogue's memory traffic
    float sink = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) sink += acc[mi][ni][0] + acc[mi][ni][15];
    if (sink == 1.2345e-30f) a.out[0] = (bf16_t)1;
    return;
  }
  const int cb0 = (cout0 + wm * (MI * 32)) / 8;
  auto swap32 = [](unsigned &lo_keeps, unsigned &hi_keeps) {  // lanes 32-63 of the first <-> lanes 0-31 of the second
    const auto r = __builtin_amdgcn_permlane32_swap(lo_keeps, hi_keeps, false, false);
    lo_keeps = r[0]; hi_keeps = r[1];
  };
  f32x4 bias[2][4];
  u32x4 rr[2][NI][2];  // residual records of channel block 4 mi + 2 gp + half, pixel tile ni
  auto preload = [&](int mi, int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cb = cb0 + mi * 4 + g;
      bias[buf][g] = cb < a.Cb_out ? *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.res) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const long long pix = p0 + wn * (NI * 32) + ni * 32 + l31;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int cb = cb0 + mi * 4 + gp * 2 + half;
          rr[buf][ni][gp] = (pix < a.P && cb < a.Cb_out) ? *reinterpret_cast<const u32x4 *>(a.res + ((size_t)cb * a.pitch_out + (size_t)pix) * 8)
                                                         : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  };
  preload(0, 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int buf = mi & 1;
    if (mi + 1 < MI) preload(mi + 1, buf ^ 1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long long pix = p0 + wn * (NI * 32) + ni * 32 + l31;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        f32x4 va, vb;  // this lane's 4 channels of blocks 2 gp and 2 gp + 1
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = acc[mi][ni][(gp * 2) * 4 + e] + bias[buf][gp * 2][e];
          vb[e] = acc[mi][ni][(gp * 2 + 1) * 4 + e] + bias[buf][gp * 2 + 1][e];
        }
        if (a.res) {
          unsigned r0 = rr[buf][ni][gp][0], r1 = rr[buf][ni][gp][1], r2 = rr[buf][ni][gp][2], r3 = rr[buf][ni][gp][3];
          swap32(r0, r2);  // lanes 0-31: r0 r1 = own low half of block 2gp, r2 r3 = low half of 2gp+1 (from lane + 32);
          swap32(r1, r3);  // lanes 32-63: r0 r1 = high half of 2gp (from lane - 32), r2 r3 = own high half of 2gp+1
          va[0] += bf2f((bf16_t)(r0 & 0xffffu)); va[1] += bf2f((bf16_t)(r0 >> 16));
          va[2] += bf2f((bf16_t)(r1 & 0xffffu)); va[3] += bf2f((bf16_t)(r1 >> 16));
          vb[0] += bf2f((bf16_t)(r2 & 0xffffu)); vb[1] += bf2f((bf16_t)(r2 >> 16));
          vb[2] += bf2f((bf16_t)(r3 & 0xffffu)); vb[3] += bf2f((bf16_t)(r3 >> 16));
        }
        {
          const int cba = cb0 + mi * 4 + gp * 2;  // va: block cba, vb: block cba + 1
          const bool rla = a.relu && !(cba >= a.norelu_cb0 && cba < a.norelu_cb1), rlb = a.relu && !(cba + 1 >= a.norelu_cb0 && cba + 1 < a.norelu_cb1);
          const float loa = relu_lo(rla), lob = relu_lo(rlb);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] = clamp_lo(va[e], loa); vb[e] = clamp_lo(vb[e], lob); }
        }
        unsigned ax = pack_bf16x2(va[0], va[1]), ay = pack_bf16x2(va[2], va[3]);
        unsigned bx = pack_bf16x2(vb[0], vb[1]), by = pack_bf16x2(vb[2], vb[3]);
        swap32(ax, bx);
        swap32(ay, by);
        const int cb = cb0 + mi * 4 + gp * 2 + half;
        if (pix < a.P && cb < a.Cb_out) *reinterpret_cast<u32x4 *>(a.out + ((size_t)cb * a.pitch_out + (size_t)pix) * 8) = u32x4{ax, ay, bx, by};
      }
    }
  }
  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.trace[2] = __builtin_amdgcn_s_memtime(); a.trace[3] = (unsigned long long)nstages; }
}

// ---- round 4: weights through LDS, PIXELS STRAIGHT INTO REGISTERS ("B-direct") ---------------------------------------------------
// What the LDS-DMA kernel above pays for, measured layer by layer with its operands knocked out (tools/bench_conv_bf16.py,
// profiles/r04_bf16_conv_ablation.txt): the pixel gather costs 16-19 % of a tower's convolution time (the weights 1 %: they are
// cache-hot), the stage barrier + fragment reads 17 %, the epilogue 5-9 %; a bare MFMA stream of its wave tile reaches 1.43 PFLOP/s
// against the 1.74 the matrix pipe sustains on random bf16 data (tools/probes/mfma_bf16_peak.cpp: power-capped).  The gather is slow
// because it is coupled: its data passes through a 2-3-stage LDS ring (72 KiB of the 80 a block may use with two blocks per CU — no
// room to run further ahead), and every wave waits at the stage barrier for the slowest wave's pixels.
// This kernel uncouples it.  Block = 128 couts x 256 pixels, the four waves side by side along the PIXEL axis: wave w owns all 128
// couts (MI = 4) of pixels 64 w .. 64 w + 63 (NI = 2).  The MFMA B fragment of a wave is then private to it — lane (l31, half) needs
// the 16-byte record of pixel l31, chunk 2 q + half: ONE buffer_load_dwordx4 per lane per fragment, coalesced (32 consecutive records),
// straight into the VGPRs the MFMA reads, no LDS, no zero-fill pass (a tap outside the map is an out-of-range buffer offset: the
// hardware returns zeros), no barrier.  The fragments of the next 2 DS - 1 k-steps (DS = 3-4 stages, 48-64 VGPRs) are in flight while
// the current one is multiplied: twice the lookahead of the ring, per wave.  Only the weights (8 KiB per stage, shared by the four
// waves) go through LDS, register-staged (two records per thread per stage) into a 3-slot ring with one barrier per stage.  Every
// load is a compiler-visible builtin: hipcc computes the vmcnt of every wait from program order, which sched_barrier pins.
// K order = (tap, chunk pair) in one accumulation chain per output, the same MFMA instruction and operand layout as the kernels
// above: results are bit-identical to theirs (tests/test_gpu_resnet.py forces either).
// ABL (timing experiments, -DMPN_BF16_ABLATE builds only; garbage results): bit 0 no pixel loads, 1 no weight loads / LDS stores,
// 2 no stage barrier, 3 no MFMAs (operands still waited for), 4 no epilogue, 5 pixel loads all from one resident 1-KiB window
// MI = 32-cout sub-tiles the block multiplies: 4, or 2 for a ragged last cout tile with <= 64 valid couts (192- / 320- / 1344-cout
// layers: their last tile no longer multiplies 64 rows of zero weights)
// PP = 1: the ping-pong form (conv2d_c8i_bf16_bdpp_kernel below): the block holds TWO such 4-wave groups, tid / wave are group-local
template <int DS, int ABL, int MI, int PP = 0>
__device__ __forceinline__ void bdir_body(const GConvArgsB &a, u32x4 (*lds_a)[4 * 128], const int tx, const int ty, const int grp = 0) {
  constexpr int NI = 2, TM = 128, TN = 256, RA = 3, RB = 2 * DS;  // RB: B-fragment ring, in k-steps
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const long long p0 = (long long)tx * TN + wave * (NI * 32);
  const int cout0 = ty * TM;
  const int OHW = a.OH * a.OW;
  const int spt = a.nch2 / 4;
  const int nstages = a.KH * a.KW * spt;
  constexpr unsigned OOB = 0x7ffffff0u;  // >= num_records of either descriptor (the host checks both tensors stay below 2 GiB)
  const unsigned plane_b = (unsigned)(a.pitch_in * 16);
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in), 0, (int)((size_t)a.nch2 * a.pitch_in * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.wpk), 0, (int)((size_t)nstages * 4 * a.CoutP * 16), 0x00020000);

  // this lane's two pixels (one per B fragment)
  int iy0[NI], ix0[NI];
  unsigned map_off[NI];
  bool pv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const long long gpix = p0 + ni * 32 + l31;
    pv[ni] = gpix < a.P;
    const int gb = pv[ni] ? (int)(gpix / OHW) : 0;
    const int grem = pv[ni] ? (int)(gpix - (long long)gb * OHW) : 0;
    const int goy = grem / a.OW, gox = grem - goy * a.OW;
    iy0[ni] = goy * a.sh - a.ph; ix0[ni] = gox * a.sw - a.pw;
    map_off[ni] = (unsigned)gb * (unsigned)(a.H * a.W);
  }
  // B issue cursor: stage, tap, channel group; per-lane byte offsets of the tap's pixel (+ this half-wave's chunk plane)
  int b_st = 0, b_cg = 0, b_kx = 0, b_ky = 0;
  unsigned voff[NI];
  auto b_offsets = [&]() {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int iy = iy0[ni] + b_ky, ix = ix0[ni] + b_kx;
      const bool ok = pv[ni] && b_st < nstages && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      voff[ni] = ok ? (map_off[ni] + (unsigned)(iy * a.W + ix)) * 16u + (unsigned)half * plane_b : OOB;
      if constexpr ((ABL & 32) != 0) voff[ni] = (unsigned)(l31 * 16 + ni * 512);
    }
  };
  b_offsets();
  bf16x8 bq[RB][NI];
  if constexpr ((ABL & 1) != 0) {
#pragma unroll
    for (int t = 0; t < RB; ++t)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bq[t][ni] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 0x3f803f80u, (unsigned)t, 0x3f803f80u});
  }
  auto b_issue = [&](int q, int slot) {  // fragments of k-step q of stage b_st
    unsigned soff = (unsigned)(b_cg * 4 + 2 * q) * plane_b;
    if constexpr ((ABL & 32) != 0) soff = 0;
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(bq[slot][ni]));
      return;
    }
    if constexpr ((ABL & 192) == 192) {  // decoupled experiment: the previous load into this slot is consumed only now
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" :: "v"(bq[slot][ni]));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[ni], soff, 0);
      bq[slot][ni] = __builtin_bit_cast(bf16x8, v);
    }
  };
  auto b_issue1 = [&](int q, int slot, int ni) {  // one fragment load (the fine-grained schedule deals them between the MFMAs)
    const unsigned soff = (unsigned)(b_cg * 4 + 2 * q) * plane_b;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[ni], soff, 0);
    bq[slot][ni] = __builtin_bit_cast(bf16x8, v);
  };
  auto b_advance = [&]() {
    ++b_st;
    if (++b_cg == spt) {
      b_cg = 0;
      if (++b_kx == a.KW) { b_kx = 0; ++b_ky; }
      b_offsets();
    } else if (b_st == nstages) {
      b_offsets();  // past the last stage (the loop runs whole groups of DS stages): zeros
    }
  };
  // A (weights): thread = row tid & 127 of chunks (tid >> 7) and (tid >> 7) + 2 of a stage; a stage past the last is an
  // out-of-range offset of the weight descriptor -> zeros
  const unsigned a_row = (unsigned)(cout0 + (tid & 127)) * 16u, a_ch = (unsigned)(tid >> 7);
  const unsigned w_chunk = (unsigned)a.CoutP * 16u;
  u32x4 a_stage[2];
  auto a_load = [&](int st) {
    if constexpr ((ABL & 2) != 0) return;
    const unsigned base = (unsigned)st * 4u * w_chunk;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = st < nstages ? a_row + (a_ch + 2u * i) * w_chunk : OOB;
      a_stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, base, 0);
    }
  };
  auto a_load1 = [&](int st, int i) {
    const unsigned base = (unsigned)st * 4u * w_chunk;  // wave-uniform
    const unsigned off = st < nstages ? a_row + (a_ch + 2u * i) * w_chunk : OOB;
    a_stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, base, 0);
  };
  auto a_store = [&](int slot) {
    if constexpr ((ABL & 2) != 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_a[slot][(a_ch + 2 * i) * TM + (tid & 127)] = a_stage[i];
  };

  bf16x8 bconst = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u ^ (unsigned)lane, 0x3f813f82u, 0x3f833f84u, 0x3f853f86u + (unsigned)lane});
  asm volatile("" : "+v"(bconst));
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  bf16x8 af[2][MI];
  auto a_frags = [&](int slot, int q, int fs) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[fs][mi] = *reinterpret_cast<const bf16x8 *>(&lds_a[slot][(2 * q + half) * TM + mi * 32 + l31]);
  };

  if constexpr (PP != 0) {
    // ---- ping-pong form.  The two groups of the block alternate, one block barrier per phase: while this group multiplies a stage
    // (phase M: 16 MFMAs back to back, nothing else), the other group — its waves share the SIMDs one to one — does everything that is
    // not an MFMA for its next stages (phase P: weight registers -> LDS, next weight loads, fragment ds_reads, the pixel loads DS
    // stages ahead, their address arithmetic), and vice versa.  Group 1 runs one phase behind group 0.
    auto phase_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    };
    // prologue (both groups at once): weights of stages 0 and 1 in LDS, of stage 2 in the staging registers; the pixel fragments of
    // stages 0 .. DS - 1 (the whole ring) in flight; the fragments of stage 0 in registers
    a_load(0);
#pragma unroll
    for (int t = 0; t < RB; ++t) {
      b_issue(t & 1, t);
      if (t & 1) b_advance();
    }
    a_store(0);
    a_load(1);
    a_store(1);
    a_load(2);
    phase_barrier();
    a_frags(0, 0, 0);
    a_frags(0, 1, 1);
    if (grp == 1) phase_barrier();
    int slot1 = 1, slot2 = 2;  // LDS slots of stages st + 1 and st + 2
    bool more = true;
    for (int g = 0; more; ++g) {
#pragma unroll
      for (int j = 0; j < DS; ++j) {
        const int st = g * DS + j;
        if (st >= nstages) { more = false; break; }
        // phase M
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int m = 0; m < MI * NI; ++m) {
            const int mi = m / NI, ni = m % NI;
            __builtin_amdgcn_sched_barrier(0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][mi], bq[2 * j + q][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        __builtin_amdgcn_s_setprio(0);
        phase_barrier();
        // phase P
        if (st + 1 < nstages) {
          a_frags(slot1, 0, 0);
          a_frags(slot1, 1, 1);
          a_store(slot2);
          a_load(st + 3);
          b_issue(0, 2 * j);
          b_issue(1, 2 * j + 1);
          b_advance();
        }
        phase_barrier();
        const int s0 = slot1 == 0 ? 2 : slot1 - 1;  // the slot of stage st: free for stage st + 3
        slot1 = slot2; slot2 = s0;
      }
    }
    if (grp == 0) phase_barrier();
  } else {
  // prologue: weights of stage 0 in LDS, of stage 1 in the staging registers; pixel fragments of k-steps 0 .. RB - 2 in flight
  a_load(0);
#pragma unroll
  for (int t = 0; t < RB - 1; ++t) {
    b_issue(t & 1, t);
    if (t & 1) b_advance();
  }
  a_store(0);
  a_load(1);
  __syncthreads();
  a_frags(0, 0, 0);

  int a_slot = 0;  // LDS slot of the stage being multiplied
  const int ngroups = (nstages + DS - 1) / DS;
  for (int g = 0; g < ngroups; ++g) {
#pragma unroll
    for (int j = 0; j < DS; ++j) {
      const int st = g * DS + j;
      const int s_nxt = a_slot + 1 == RA ? 0 : a_slot + 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 2 * j + q;                         // k-step within the group = its ring slot (static)
        const int t_iss = (t + RB - 1) % RB;             // slot freed by the previous k-step: k-step t + RB - 1 goes there
        // issue side: the pixel fragments RB - 1 k-steps ahead (cursor parity: the prologue left it at k-step RB - 1 = odd)
        constexpr bool FINE = (ABL & 0x1ff) == 0 || (ABL & 512) != 0;  // (the knock-out experiments keep the clumped schedule they were measured on)
        // FINE: every load of the k-step is issued BETWEEN its MFMAs, two MFMAs apart, instead of in a clump ahead of them.  Measured
        // on a bare loop (tools/probes/vmem_path_bw.cpp: 3 x 1-KiB loads per 8 MFMAs, L2-resident): clumped 0.97 PFLOP/s, one load
        // every 2-3 MFMAs 1.48 — a clump from all eight waves of the CU at once backs up the vector-memory issue path and the waves
        // stand at their loads instead of their MFMAs.
        constexpr int T = MI * NI;
        constexpr int PB0 = T == 8 ? 1 : 0, PB1 = T == 8 ? 3 : 1, PA0 = T == 8 ? 5 : 2, PA1 = T == 8 ? 6 : 3;  // the MFMA each load follows
        if constexpr (!FINE) {
          b_issue((q + 1) & 1, t_iss);
          if (((q + 1) & 1) == 1) b_advance();
        }
        if (q == 0) {  // weights: stage st + 1 from the staging registers into its slot (last read two stages ago), stage st + 2 into the registers
          if constexpr (!FINE) { a_store(s_nxt); a_load(st + 2); }
        }
        if (q == 1) {  // the stage barrier sits before the last k-step's MFMAs (their operands are in registers already)
          if constexpr ((ABL & 4) == 0) __syncthreads();
          a_frags(s_nxt, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MI * NI; ++m) {
          const int mi = m / NI, ni = m % NI;
          if constexpr (FINE) {
            if (m == PB0 + 1) b_issue1((q + 1) & 1, t_iss, 0);
            if (m == PB1 + 1) { b_issue1((q + 1) & 1, t_iss, 1); if (((q + 1) & 1) == 1) b_advance(); if (q == 0) a_store(s_nxt); }
            if (q == 0 && m == PA0 + 1) a_load1(st + 2, 0);
            if (q == 0 && m == PA1 + 1 && PA1 + 1 < T) a_load1(st + 2, 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr ((ABL & 8) != 0) asm volatile("" :: "v"(af[q][mi]), "v"(bq[t][ni]));
          else if constexpr ((ABL & 64) != 0) {  // MFMAs on a constant pixel fragment: the loads run beside them without feeding them
            if constexpr ((ABL & 128) == 0) { if (mi == 0) asm volatile("" :: "v"(bq[t][ni])); }
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][mi], bconst, acc[mi][ni], 0, 0, 0);
          } else
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][mi], bq[t][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (q == 0 && m == 0) a_frags(a_slot, 1, 1);
          if constexpr (FINE) { if (q == 0 && m == T - 1 && PA1 + 1 >= T) a_load1(st + 2, 1); }
        }
      }
      a_slot = s_nxt;
    }
  }
  }  // PP

  if constexpr ((ABL & 16) != 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
    return;
  }
  // epilogue: as the LDS-DMA kernel's (loads of channel group mi + 1 before the stores of group mi; 16-byte accesses through
  // v_permlane32_swap), for this wave's 128 couts x 64 pixels
  const int cb0 = cout0 / 8;
  auto swap32 = [](unsigned &lo_keeps, unsigned &hi_keeps) {
    const auto r = __builtin_amdgcn_permlane32_swap(lo_keeps, hi_keeps, false, false);
    lo_keeps = r[0]; hi_keeps = r[1];
  };
  f32x4 bias[2][4];
  u32x4 rr[2][NI][2];
  auto preload = [&](int mi, int buf) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int cb = cb0 + mi * 4 + gq;
      bias[buf][gq] = cb < a.Cb_out ? *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.res) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const long long pix = p0 + ni * 32 + l31;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int cb = cb0 + mi * 4 + gp * 2 + half;
          rr[buf][ni][gp] = (pix < a.P && cb < a.Cb_out) ? *reinterpret_cast<const u32x4 *>(a.res + ((size_t)cb * a.pitch_out + (size_t)pix) * 8)
                                                         : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  };
  preload(0, 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int buf = mi & 1;
    if (mi + 1 < MI) preload(mi + 1, buf ^ 1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long long pix = p0 + ni * 32 + l31;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        f32x4 va, vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = acc[mi][ni][(gp * 2) * 4 + e] + bias[buf][gp * 2][e];
          vb[e] = acc[mi][ni][(gp * 2 + 1) * 4 + e] + bias[buf][gp * 2 + 1][e];
        }
        if (a.res) {
          unsigned r0 = rr[buf][ni][gp][0], r1 = rr[buf][ni][gp][1], r2 = rr[buf][ni][gp][2], r3 = rr[buf][ni][gp][3];
          swap32(r0, r2);
          swap32(r1, r3);
          va[0] += bf2f((bf16_t)(r0 & 0xffffu)); va[1] += bf2f((bf16_t)(r0 >> 16));
          va[2] += bf2f((bf16_t)(r1 & 0xffffu)); va[3] += bf2f((bf16_t)(r1 >> 16));
          vb[0] += bf2f((bf16_t)(r2 & 0xffffu)); vb[1] += bf2f((bf16_t)(r2 >> 16));
          vb[2] += bf2f((bf16_t)(r3 & 0xffffu)); vb[3] += bf2f((bf16_t)(r3 >> 16));
        }
        {
          const int cba = cb0 + mi * 4 + gp * 2;
          const bool rla = a.relu && !(cba >= a.norelu_cb0 && cba < a.norelu_cb1), rlb = a.relu && !(cba + 1 >= a.norelu_cb0 && cba + 1 < a.norelu_cb1);
          const float loa = relu_lo(rla), lob = relu_lo(rlb);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] = clamp_lo(va[e], loa); vb[e] = clamp_lo(vb[e], lob); }
        }
        unsigned ax = pack_bf16x2(va[0], va[1]), ay = pack_bf16x2(va[2], va[3]);
        unsigned bx = pack_bf16x2(vb[0], vb[1]), by = pack_bf16x2(vb[2], vb[3]);
        swap32(ax, bx);
        swap32(ay, by);
        const int cb = cb0 + mi * 4 + gp * 2 + half;
        if (pix < a.P && cb < a.Cb_out) *reinterpret_cast<u32x4 *>(a.out + ((size_t)cb * a.pitch_out + (size_t)pix) * 8) = u32x4{ax, ay, bx, by};
      }
    }
  }
}

template <int DS, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv2d_c8i_bf16_bdir_kernel(GConvArgsB a, int nx, int ny) {
  __shared__ __attribute__((aligned(16))) u32x4 lds_a[(ABL & 8) ? 9 : 3][4 * 128];  // [slot][chunk][cout row] 16-byte records (no-MFMA timing variant: padded so that still two blocks fit a CU)
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;  // a pixel tile's cout tiles back to back on one XCD
  if (tx >= nx) return;
  if (a.Cb_out * 8 - ty * 128 > 64) bdir_body<DS, ABL, 4>(a, lds_a, tx, ty);
  else bdir_body<DS, ABL, 2>(a, lds_a, tx, ty);
}

#ifdef MPN_DEBUG_HOOKS  // measured 13 % SLOWER than two free-running blocks per CU: debug flavour only (profiles/r04_bf16_conv_bdir_ablation.txt, F)
// The ping-pong form: 512 threads = two 4-wave groups, each a B-direct block of its own (own tile, own weight ring), phase-locked
// by the block barrier so that one group's MFMA phase always runs beside the other group's memory phase (bdir_body<.., PP = 1>).
// Tiles: virtual block v = 2 * (blockIdx.x >> 3) + group on XCD blockIdx.x & 7 — the two groups normally hold neighbouring cout
// tiles of ONE pixel tile (the second group's pixel loads hit the lines the first group just fetched).
// Result: bit-identical, 4.65 vs 4.11 ms per Inception tower at a HIGHER clock (2.35 vs 2.2 GHz: the matrix pipe idles more) — a
// phase lasts as long as the memory path needs for a stage's 24 KiB, not as long as its 16 MFMAs: the vector-memory path
// (~30 B / clock / CU measured with the loads alone) is the limit, and lock-stepping the groups only adds the barrier skew.
template <int DS>
__global__ __launch_bounds__(512, 1) void conv2d_c8i_bf16_bdpp_kernel(GConvArgsB a, int nx, int ny) {
  __shared__ __attribute__((aligned(16))) u32x4 lds_a[2][3][4 * 128];  // [group][slot][chunk][cout row] 16-byte records
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int xcd = blockIdx.x & 7, kq = (int)(blockIdx.x >> 3) * 2 + grp;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;
  if (tx >= nx) {  // this group has no tile: keep the other group's barrier count
    const int nb = 2 + 2 * (a.KH * a.KW * (a.nch2 / 4));
    for (int i = 0; i < nb; ++i) __syncthreads();
    return;
  }
  if (a.Cb_out * 8 - ty * 128 > 64) bdir_body<DS, 0, 4, 1>(a, lds_a[grp], tx, ty, grp);
  else bdir_body<DS, 0, 2, 1>(a, lds_a[grp], tx, ty, grp);
}
#endif  // MPN_DEBUG_HOOKS (ping-pong form)

#ifdef MPN_DEBUG_HOOKS  // measured 12 % SLOWER than the 4-wave kernel above: debug flavour only (profiles/r04_bf16_conv_bdir_ablation.txt)
// ---- round 4, an experiment that did not pay: B-direct with 256 couts per block ("bdir8") ---------------------------------------
// Knock-out timing of the kernel above (tools/bench_conv_bf16.py bf16_bdir_abl=..., profiles/r04_bf16_conv_bdir_ablation.txt), per
// Inception tower, sustained: MFMAs alone 1.84-2.28 ms, the loads alone 2.47, everything 4.12 — the matrix stream and the vector-memory
// stream add up instead of overlapping, and pixel loads that all HIT the CU's vector cache still cost half of what the real ones do.
// Hypothesis tested here: fewer bytes per FLOP.  256 couts per block, eight waves side by side along the PIXEL axis, each owning ALL
// 256 couts (MI = 8) of 32 pixels (NI = 1): a pixel fragment feeds eight MFMAs instead of four (16 KiB weights + 16 KiB pixels per
// 4.2 MFLOP stage against 8 + 16 per 2.1).  One block per CU (two waves per SIMD as before), 128 accumulator registers; the weight
// fragments are single-buffered and ROLL (fragment mi of the next k-step is read right behind the MFMA that consumed fragment mi of
// this one); pixel ring of 2 DS k-steps (DS = 4, 6, 8 instantiated: 220-252 VGPRs, no spill).  Bit-identical to the kernels above.
// Result: slower on every 384- / 1344-cout layer (its MFMA stream with one LDS read per MFMA runs at 3.26 vs 2.78 ms per tower even
// with no loads at all), a deeper ring changes nothing (4.69 / 4.70 / 4.75 ms for 8 / 12 / 16 k-steps in flight): neither bytes per
// FLOP nor lookahead is what keeps the two streams from overlapping.  Kept for the record and for tools/bench_conv_bf16.py.
// MI = the block's valid 32-cout sub-tiles rounded up to even (compile-time: the kernel dispatches once per block)
template <int DS, int MI, int ABL>
__device__ __forceinline__ void bdir8_body(const GConvArgsB &a, u32x4 (*lds_a)[4 * 256], const int tx, const int ty) {
  constexpr int TM = 256, TN = 256, RA = 3, RB = 2 * DS;  // RB: pixel-fragment ring, in k-steps
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const long long p0 = (long long)tx * TN + wave * 32;
  const int cout0 = ty * TM;
  const int OHW = a.OH * a.OW;
  const int spt = a.nch2 / 4;
  const int nstages = a.KH * a.KW * spt;
  constexpr unsigned OOB = 0x7ffffff0u;
  const unsigned plane_b = (unsigned)(a.pitch_in * 16);
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.in), 0, (int)((size_t)a.nch2 * a.pitch_in * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(a.wpk), 0, (int)((size_t)nstages * 4 * a.CoutP * 16), 0x00020000);

  // this lane's pixel
  int iy0, ix0;
  unsigned map_off;
  bool pv;
  {
    const long long gpix = p0 + l31;
    pv = gpix < a.P;
    const int gb = pv ? (int)(gpix / OHW) : 0;
    const int grem = pv ? (int)(gpix - (long long)gb * OHW) : 0;
    const int goy = grem / a.OW, gox = grem - goy * a.OW;
    iy0 = goy * a.sh - a.ph; ix0 = gox * a.sw - a.pw;
    map_off = (unsigned)gb * (unsigned)(a.H * a.W);
  }
  int b_st = 0, b_cg = 0, b_kx = 0, b_ky = 0;
  unsigned voff;
  auto b_offsets = [&]() {
    const int iy = iy0 + b_ky, ix = ix0 + b_kx;
    const bool ok = pv && b_st < nstages && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    voff = ok ? (map_off + (unsigned)(iy * a.W + ix)) * 16u + (unsigned)half * plane_b : OOB;
    if constexpr ((ABL & 32) != 0) voff = (unsigned)(l31 * 16);
  };
  b_offsets();
  bf16x8 bq[RB];
  if constexpr ((ABL & 1) != 0) {
#pragma unroll
    for (int t = 0; t < RB; ++t) bq[t] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, 0x3f803f80u, (unsigned)t, 0x3f803f80u});
  }
  auto b_issue = [&](int q, int slot) {  // the fragment of k-step q of stage b_st
    unsigned soff = (unsigned)(b_cg * 4 + 2 * q) * plane_b;
    if constexpr ((ABL & 32) != 0) soff = 0;
    if constexpr ((ABL & 1) != 0) {
      asm volatile("" : "+v"(bq[slot]));
      return;
    }
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, soff, 0);
    bq[slot] = __builtin_bit_cast(bf16x8, v);
  };
  auto b_advance = [&]() {
    ++b_st;
    if (++b_cg == spt) {
      b_cg = 0;
      if (++b_kx == a.KW) { b_kx = 0; ++b_ky; }
      b_offsets();
    } else if (b_st == nstages) {
      b_offsets();  // past the last stage: zeros
    }
  };
  // A (weights): thread = row tid & 255 of chunks (tid >> 8) and (tid >> 8) + 2 of a stage; rows past the padded cout count and
  // stages past the last are out-of-range offsets -> zeros
  const bool a_ok = cout0 + (tid & 255) < a.CoutP;
  const unsigned a_row = (unsigned)(cout0 + (tid & 255)) * 16u, a_ch = (unsigned)(tid >> 8);
  const unsigned w_chunk = (unsigned)a.CoutP * 16u;
  u32x4 a_stage[2];
  auto a_load = [&](int st) {
    if constexpr ((ABL & 2) != 0) return;
    const unsigned base = (unsigned)st * 4u * w_chunk;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = (a_ok && st < nstages) ? a_row + (a_ch + 2u * i) * w_chunk : OOB;
      a_stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, off, base, 0);
    }
  };
  auto a_store = [&](int slot) {
    if constexpr ((ABL & 2) != 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_a[slot][(a_ch + 2 * i) * TM + (tid & 255)] = a_stage[i];
  };

  f32x16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
  bf16x8 af[MI];
  auto a_frag = [&](int slot, int q, int mi) { af[mi] = *reinterpret_cast<const bf16x8 *>(&lds_a[slot][(2 * q + half) * TM + mi * 32 + l31]); };

  // prologue: weights of stage 0 in LDS, of stage 1 in the staging registers; pixel fragments of k-steps 0 .. RB - 2 in flight
  a_load(0);
#pragma unroll
  for (int t = 0; t < RB - 1; ++t) {
    b_issue(t & 1, t);
    if (t & 1) b_advance();
  }
  a_store(0);
  a_load(1);
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a_frag(0, 0, mi);

  int a_slot = 0;  // LDS slot of the stage being multiplied
  bool more = true;
  for (int g = 0; more; ++g) {
#pragma unroll
    for (int j = 0; j < DS; ++j) {
      const int st = g * DS + j;
      if (st >= nstages) { more = false; break; }
      const int s_nxt = a_slot + 1 == RA ? 0 : a_slot + 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 2 * j + q;              // k-step within the group = its ring slot (static)
        const int t_iss = (t + RB - 1) % RB;  // the slot the previous k-step freed: k-step t + RB - 1 goes there
        b_issue((q + 1) & 1, t_iss);
        if (((q + 1) & 1) == 1) b_advance();
        if (q == 0) {  // weights: stage st + 1 from the staging registers into its slot (last read two stages ago), stage st + 2 into the registers
          a_store(s_nxt);
          a_load(st + 2);
        }
        if (q == 1) {  // the stage barrier: from here on the rolling reads fetch stage st + 1's fragments
          if constexpr ((ABL & 4) == 0) __syncthreads();
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr ((ABL & 8) != 0) asm volatile("" :: "v"(af[mi]), "v"(bq[t]));
          else
          acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bq[t], acc[mi], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (q == 0) a_frag(a_slot, 1, mi); else a_frag(s_nxt, 0, mi);
        }
      }
      a_slot = s_nxt;
    }
  }

  if constexpr ((ABL & 16) != 0) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) asm volatile("" :: "v"(acc[mi]));
    return;
  }
  // epilogue: as the kernels above (loads of channel group mi + 1 before the stores of group mi; 16-byte accesses through
  // v_permlane32_swap), for this wave's 256 couts x 32 pixels
  const int cb0 = cout0 / 8;
  auto swap32 = [](unsigned &lo_keeps, unsigned &hi_keeps) {
    const auto r = __builtin_amdgcn_permlane32_swap(lo_keeps, hi_keeps, false, false);
    lo_keeps = r[0]; hi_keeps = r[1];
  };
  const long long pix = p0 + l31;
  f32x4 bias[2][4];
  u32x4 rr[2][2];
  auto preload = [&](int mi, int buf) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int cb = cb0 + mi * 4 + gq;
      bias[buf][gq] = cb < a.Cb_out ? *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.res) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int cb = cb0 + mi * 4 + gp * 2 + half;
        rr[buf][gp] = (pix < a.P && cb < a.Cb_out) ? *reinterpret_cast<const u32x4 *>(a.res + ((size_t)cb * a.pitch_out + (size_t)pix) * 8)
                                                   : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  preload(0, 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int buf = mi & 1;
    if (mi + 1 < MI) preload(mi + 1, buf ^ 1);
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      f32x4 va, vb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = acc[mi][(gp * 2) * 4 + e] + bias[buf][gp * 2][e];
        vb[e] = acc[mi][(gp * 2 + 1) * 4 + e] + bias[buf][gp * 2 + 1][e];
      }
      if (a.res) {
        unsigned r0 = rr[buf][gp][0], r1 = rr[buf][gp][1], r2 = rr[buf][gp][2], r3 = rr[buf][gp][3];
        swap32(r0, r2);
        swap32(r1, r3);
        va[0] += bf2f((bf16_t)(r0 & 0xffffu)); va[1] += bf2f((bf16_t)(r0 >> 16));
        va[2] += bf2f((bf16_t)(r1 & 0xffffu)); va[3] += bf2f((bf16_t)(r1 >> 16));
        vb[0] += bf2f((bf16_t)(r2 & 0xffffu)); vb[1] += bf2f((bf16_t)(r2 >> 16));
        vb[2] += bf2f((bf16_t)(r3 & 0xffffu)); vb[3] += bf2f((bf16_t)(r3 >> 16));
      }
      {
        const int cba = cb0 + mi * 4 + gp * 2;
        const bool rla = a.relu && !(cba >= a.norelu_cb0 && cba < a.norelu_cb1), rlb = a.relu && !(cba + 1 >= a.norelu_cb0 && cba + 1 < a.norelu_cb1);
        const float loa = relu_lo(rla), lob = relu_lo(rlb);
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = clamp_lo(va[e], loa); vb[e] = clamp_lo(vb[e], lob); }
      }
      unsigned ax = pack_bf16x2(va[0], va[1]), ay = pack_bf16x2(va[2], va[3]);
      unsigned bx = pack_bf16x2(vb[0], vb[1]), by = pack_bf16x2(vb[2], vb[3]);
      swap32(ax, bx);
      swap32(ay, by);
      const int cb = cb0 + mi * 4 + gp * 2 + half;
      if (pix < a.P && cb < a.Cb_out) *reinterpret_cast<u32x4 *>(a.out + ((size_t)cb * a.pitch_out + (size_t)pix) * 8) = u32x4{ax, ay, bx, by};
    }
  }
}

template <int DS, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv2d_c8i_bf16_bdir8_kernel(GConvArgsB a, int nx, int ny) {
  __shared__ __attribute__((aligned(16))) u32x4 lds_a[3][4 * 256];  // [slot][chunk][cout row] 16-byte records
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;  // a pixel tile's cout tiles back to back on one XCD
  if (tx >= nx) return;
  const int mi_cnt = (a.Cb_out * 8 - ty * 256 + 31) / 32;  // valid 32-cout sub-tiles of this block (block-uniform)
  if (mi_cnt > 6) bdir8_body<DS, 8, ABL>(a, lds_a, tx, ty);
  else if (mi_cnt > 4) bdir8_body<DS, 6, ABL>(a, lds_a, tx, ty);
  else if (mi_cnt > 2) bdir8_body<DS, 4, ABL>(a, lds_a, tx, ty);
  else bdir8_body<DS, 2, ABL>(a, lds_a, tx, ty);
}

#endif  // MPN_DEBUG_HOOKS (bdir8)

#ifdef MPN_DEBUG_HOOKS  // measured no faster than the compiler-counted form (4.73-4.78 vs 4.73 ms per Inception tower): kept in the debug flavour only
// Second form of the same kernel: every memory operation of the K loop is issued by inline asm and counted by hand.  Why: vmcnt retires
// IN ORDER, so in the form above the wait for a stage's weight registers (loaded one stage earlier) also waits for every pixel fragment
// issued before them — the pixel lookahead the ring was built for is cut from 2 DS - 1 k-steps to about three.  Here the weights go
// global -> LDS by LDS-DMA (no staging registers, no ds_write) into a SIX-slot ring, issued four stages ahead: by the time a stage's
// weights are waited for (vmcnt(18), before the stage barrier) every older pixel fragment has long been consumed, so the wait costs the
// pixel stream nothing.  Every k-step issues exactly [fragment, fragment, weight piece]; a fragment pair is therefore followed by
// 3 (RB - 1) + 1 operations when its k-step comes up: s_waitcnt vmcnt(3 RB - 2), with the fragment registers as in/out operands of the
// wait statement so that no MFMA is scheduled above it.  Past the last stage the fragment loads are out-of-range offsets (zeros) and the
// weight pieces re-load stage 0 (finite values x 0).
template <int DS>
__global__ __launch_bounds__(256, 2) void conv2d_c8i_bf16_bdir2_kernel(GConvArgsB a, int nx, int ny) {
  constexpr int MI = 4, NI = 2, TM = 128, TN = 256, RA = 6, LA = 4, RB = 2 * DS;  // RA / LA: weight ring slots / stages of weight lookahead; RB: fragment ring (k-steps)
  __shared__ __attribute__((aligned(16))) u32x4 lds_a[RA][4 * TM];                // [slot][chunk][cout row] 16-byte records
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;  // a pixel tile's cout tiles back to back on one XCD
  if (tx >= nx) return;
  const long long p0 = (long long)tx * TN + wave * (NI * 32);
  const int cout0 = ty * TM;
  const int OHW = a.OH * a.OW;
  const int spt = a.nch2 / 4;
  const int nstages = a.KH * a.KW * spt;
  constexpr unsigned OOB = 0x7ffffff0u;  // >= num_records of either descriptor (the host checks both tensors stay below 2 GiB)
  const unsigned plane_b = (unsigned)(a.pitch_in * 16);
  u32x4 rs_in;  // raw buffer descriptor of the input tensor in four SGPRs: base, stride 0, num_records (bytes), flags
  {
    const unsigned long long base = (unsigned long long)a.in;
    rs_in[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    rs_in[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
    rs_in[2] = __builtin_amdgcn_readfirstlane((unsigned)((size_t)a.nch2 * a.pitch_in * 16));
    rs_in[3] = __builtin_amdgcn_readfirstlane(0x00020000u);
  }

  // this lane's two pixels (one per B fragment)
  int iy0[NI], ix0[NI];
  unsigned map_off[NI];
  bool pv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const long long gpix = p0 + ni * 32 + l31;
    pv[ni] = gpix < a.P;
    const int gb = pv[ni] ? (int)(gpix / OHW) : 0;
    const int grem = pv[ni] ? (int)(gpix - (long long)gb * OHW) : 0;
    const int goy = grem / a.OW, gox = grem - goy * a.OW;
    iy0[ni] = goy * a.sh - a.ph; ix0[ni] = gox * a.sw - a.pw;
    map_off[ni] = (unsigned)gb * (unsigned)(a.H * a.W);
  }
  // B issue cursor: stage, tap, channel group; per-lane byte offsets of the tap's pixel (+ this half-wave's chunk plane)
  int b_st = 0, b_cg = 0, b_kx = 0, b_ky = 0;
  unsigned voff[NI];
  auto b_offsets = [&]() {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int iy = iy0[ni] + b_ky, ix = ix0[ni] + b_kx;
      const bool ok = pv[ni] && b_st < nstages && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      voff[ni] = ok ? (map_off[ni] + (unsigned)(iy * a.W + ix)) * 16u + (unsigned)half * plane_b : OOB;
    }
  };
  b_offsets();
  bf16x8 bq[RB][NI];
  auto b_issue = [&](int q, int slot) {  // fragments of k-step q of stage b_st (destinations unprotected until the counted wait)
    const unsigned soff = (unsigned)(b_cg * 4 + 2 * q) * plane_b;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[slot][ni]) : "v"(voff[ni]), "s"(rs_in), "s"(soff) : "memory");
  };
  auto b_advance = [&]() {
    ++b_st;
    if (++b_cg == spt) {
      b_cg = 0;
      if (++b_kx == a.KW) { b_kx = 0; ++b_ky; }
      b_offsets();
    } else if (b_st == nstages) {
      b_offsets();  // past the last stage (the loop runs whole groups of DS stages): zeros
    }
  };
  // A (weights) by LDS-DMA: a stage is 4 chunks x 128 rows = 8 wave-loads of 1 KiB; wave w moves pieces w (k-step 0) and w + 4 (k-step 1):
  // piece p = rows 64 (p & 1) .. of chunk p >> 1.  Stage st + LA goes to slot (st + LA) % RA while stage st is multiplied.
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const u32x4 *)&lds_a[0][0]);
  const unsigned a_lane = (unsigned)lane * 16u;
  const size_t w_chunk = (size_t)a.CoutP * 16;
  const char *const w_base = reinterpret_cast<const char *>(a.wpk) + (size_t)cout0 * 16;
  int a_st = 0, a_slot_iss = 0;  // issue cursor: stage, its ring slot
  auto a_issue = [&](int half_piece) {  // half_piece 0 / 1 = this wave's first / second piece of stage a_st
    const int p = wave + 4 * half_piece;
    const int st_src = a_st < nstages ? a_st : 0;  // past the end: any finite weights (their pixel fragments are zeros)
    const char *src = w_base + ((size_t)st_src * 4 + (p >> 1)) * w_chunk + (size_t)(p & 1) * 1024;
    glds16_s(src, a_lane, lds0 + (unsigned)a_slot_iss * (4u * TM * 16u) + (unsigned)((p >> 1) * TM + (p & 1) * 64) * 16u);
  };
  auto a_advance = [&]() { ++a_st; a_slot_iss = a_slot_iss + 1 == RA ? 0 : a_slot_iss + 1; };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
  bf16x8 af[2][MI];
  auto a_frags = [&](int slot, int q, int fs) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[fs][mi] = *reinterpret_cast<const bf16x8 *>(&lds_a[slot][(2 * q + half) * TM + mi * 32 + l31]);
  };

  // prologue: weight stages 0 .. LA - 1 and the pixel fragments of k-steps 0 .. RB - 2 in flight, then ONE full drain (every later
  // wait is counted against the steady [fragment, fragment, weight piece] pattern, which starts after this point)
#pragma unroll
  for (int i = 0; i < LA; ++i) { a_issue(0); a_issue(1); a_advance(); }
#pragma unroll
  for (int t = 0; t < RB - 1; ++t) {
    b_issue(t & 1, t);
    if (t & 1) b_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  a_frags(0, 0, 0);

  int a_slot = 0;  // LDS slot of the stage being multiplied
  const int ngroups = (nstages + DS - 1) / DS;
  for (int g = 0; g < ngroups; ++g) {
#pragma unroll
    for (int j = 0; j < DS; ++j) {
      const int s_nxt = a_slot + 1 == RA ? 0 : a_slot + 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 2 * j + q;                         // k-step within the group = its ring slot (static)
        const int t_iss = (t + RB - 1) % RB;             // slot freed by the previous k-step: k-step t + RB - 1 goes there
        // issue side, exactly three operations per k-step: the two pixel fragments RB - 1 k-steps ahead, one weight piece LA stages ahead
        b_issue((q + 1) & 1, t_iss);
        if (((q + 1) & 1) == 1) b_advance();
        a_issue(q);
        if (q == 1) a_advance();
        if (q == 1) {  // the stage barrier sits before the last k-step's MFMAs (their operands are in registers already)
          // the next stage's weight pieces were issued LA - 1 stages ago: 3 operations per k-step since, this k-step's included
          static_assert(LA == 4, "the counted wait below assumes the weight pieces of stage st + 1 were issued during stage st - 3");
          asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          __syncthreads();
          a_frags(s_nxt, 0, 0);
        }
        // this k-step's fragments: issued RB - 1 k-steps ago, followed since by 1 + 3 (RB - 2) + 3 operations
        {
          constexpr int NW = 3 * RB - 2;
          static_assert(NW == 16 || NW == 22, "");
          if constexpr (NW == 16) asm volatile("s_waitcnt vmcnt(16)" : "+v"(bq[t][0]), "+v"(bq[t][1]) :: "memory");
          else asm volatile("s_waitcnt vmcnt(22)" : "+v"(bq[t][0]), "+v"(bq[t][1]) :: "memory");
        }
#pragma unroll
        for (int m = 0; m < MI * NI; ++m) {
          const int mi = m / NI, ni = m % NI;
          __builtin_amdgcn_sched_barrier(0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q][mi], bq[t][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (q == 0 && m == 0) a_frags(a_slot, 1, 1);
        }
      }
      a_slot = s_nxt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead loads past the last stage (zeros / stage-0 weights): nothing may land after the wave has gone

  // epilogue: as the LDS-DMA kernel's (loads of channel group mi + 1 before the stores of group mi; 16-byte accesses through
  // v_permlane32_swap), for this wave's 128 couts x 64 pixels
  const int cb0 = cout0 / 8;
  auto swap32 = [](unsigned &lo_keeps, unsigned &hi_keeps) {
    const auto r = __builtin_amdgcn_permlane32_swap(lo_keeps, hi_keeps, false, false);
    lo_keeps = r[0]; hi_keeps = r[1];
  };
  f32x4 bias[2][4];
  u32x4 rr[2][NI][2];
  auto preload = [&](int mi, int buf) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int cb = cb0 + mi * 4 + gq;
      bias[buf][gq] = cb < a.Cb_out ? *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.res) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const long long pix = p0 + ni * 32 + l31;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int cb = cb0 + mi * 4 + gp * 2 + half;
          rr[buf][ni][gp] = (pix < a.P && cb < a.Cb_out) ? *reinterpret_cast<const u32x4 *>(a.res + ((size_t)cb * a.pitch_out + (size_t)pix) * 8)
                                                         : u32x4{0u, 0u, 0u, 0u};
        }
      }
    }
  };
  preload(0, 0);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int buf = mi & 1;
    if (mi + 1 < MI) preload(mi + 1, buf ^ 1);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const long long pix = p0 + ni * 32 + l31;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        f32x4 va, vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = acc[mi][ni][(gp * 2) * 4 + e] + bias[buf][gp * 2][e];
          vb[e] = acc[mi][ni][(gp * 2 + 1) * 4 + e] + bias[buf][gp * 2 + 1][e];
        }
        if (a.res) {
          unsigned r0 = rr[buf][ni][gp][0], r1 = rr[buf][ni][gp][1], r2 = rr[buf][ni][gp][2], r3 = rr[buf][ni][gp][3];
          swap32(r0, r2);
          swap32(r1, r3);
          va[0] += bf2f((bf16_t)(r0 & 0xffffu)); va[1] += bf2f((bf16_t)(r0 >> 16));
          va[2] += bf2f((bf16_t)(r1 & 0xffffu)); va[3] += bf2f((bf16_t)(r1 >> 16));
          vb[0] += bf2f((bf16_t)(r2 & 0xffffu)); vb[1] += bf2f((bf16_t)(r2 >> 16));
          vb[2] += bf2f((bf16_t)(r3 & 0xffffu)); vb[3] += bf2f((bf16_t)(r3 >> 16));
        }
        {
          const int cba = cb0 + mi * 4 + gp * 2;
          const bool rla = a.relu && !(cba >= a.norelu_cb0 && cba < a.norelu_cb1), rlb = a.relu && !(cba + 1 >= a.norelu_cb0 && cba + 1 < a.norelu_cb1);
          const float loa = relu_lo(rla), lob = relu_lo(rlb);
#pragma unroll
          for (int e = 0; e < 4; ++e) { va[e] = clamp_lo(va[e], loa); vb[e] = clamp_lo(vb[e], lob); }
        }
        unsigned ax = pack_bf16x2(va[0], va[1]), ay = pack_bf16x2(va[2], va[3]);
        unsigned bx = pack_bf16x2(vb[0], vb[1]), by = pack_bf16x2(vb[2], vb[3]);
        swap32(ax, bx);
        swap32(ay, by);
        const int cb = cb0 + mi * 4 + gp * 2 + half;
        if (pix < a.P && cb < a.Cb_out) *reinterpret_cast<u32x4 *>(a.out + ((size_t)cb * a.pitch_out + (size_t)pix) * 8) = u32x4{ax, ay, bx, by};
      }
    }
  }
}

#endif  // MPN_DEBUG_HOOKS (conv2d_c8i_bf16_bdir2_kernel)

// ---- fp32 generic convolution, LDS-DMA + hand-pipelined form (32-channel stages) ------------------------------------------
// Same tile and arithmetic as conv2d_c8i_kernel<4> (128 couts x 128 pixels, 4 waves of 64 x 64, fp32 MFMA, stage = 32 input
// channels of one tap, two stages in LDS), but built like dense.hip's gemm_c8_pf_kernel: both operands go global -> LDS through
// global_load_lds (SADDR form; the weights are a linear copy — consecutive stages are consecutive memory in
// [tap][chunk][CoutP][8] — and the im2col gather is `chunk plane + the lane's pixel offset`), no staging registers or ds_write;
// fragments are double-buffered one 8-channel chunk ahead, each chunk's first MFMA is issued before the next chunk's fragment
// reads, the next stage's 8 DMA items follow MFMAs 1..4 of the first chunk, and the stage barrier sits before the LAST chunk's
// MFMAs.  Thread = (row tid >> 1, 16-byte half tid & 1) of the 128-row slices, i.e. DMA piece wave * 64 + lane of each chunk.  A
// lane whose tap falls outside its map loads its map's first pixel and overwrites the four records with zeros once they landed.
__global__ __launch_bounds__(256) void conv2d_c8i_pf_kernel(GConvArgs a) {
  constexpr int KCH = 4, OP_FLOATS = KCH * 128 * 8, STAGE = 2 * OP_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float ldsf[];  // [2 stages][A | B][chunk][128 rows][8]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int ny = a.CoutP / 128, nx = (int)((a.P + 127) / 128);
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int ty = kq % ny, tx = (kq / ny) * 8 + xcd;
  if (tx >= nx) return;
  const long long p0 = (long long)tx * 128;
  const int cout0 = ty * 128;
  const int OHW = a.OH * a.OW;
  const int srow = tid >> 1, sh = tid & 1;
  const long long gpix = p0 + srow;
  const bool gvalid = gpix < a.P;
  const int gb = gvalid ? (int)(gpix / OHW) : 0;
  const int grem = gvalid ? (int)(gpix - (long long)gb * OHW) : 0;
  const int goy = grem / a.OW, gox = grem - goy * a.OW;
  const int iy0 = goy * a.sh - a.ph, ix0 = gox * a.sw - a.pw;
  const unsigned map_off = (unsigned)gb * (unsigned)(a.H * a.W);  // records of 32 bytes

  const int spt = a.nch / KCH;
  const int nstages = a.KH * a.KW * spt;
  const int st0 = a.part ? (int)blockIdx.z * a.stages_per_split : 0;
  const int st1 = a.part ? min(nstages, st0 + a.stages_per_split) : nstages;
  if (st0 >= st1) return;  // (never: every split owns at least one stage)

  // issue cursor, positioned at stage st0
  int i_tap = st0 / spt, i_cg = st0 - i_tap * spt;
  int i_ky = i_tap / a.KW, i_kx = i_tap - i_ky * a.KW;
  const size_t w_step = (size_t)a.CoutP * 32, b_step = a.pitch_in * 32;  // bytes per chunk
  const char *i_wp = reinterpret_cast<const char *>(a.wpk + (size_t)cout0 * 8) + ((size_t)i_tap * a.nch + (size_t)i_cg * KCH) * w_step;
  const char *const b_base = reinterpret_cast<const char *>(a.in);
  const char *i_bp = b_base + (size_t)i_cg * KCH * b_step;
  const unsigned a_lane = (unsigned)tid * 16u;
  unsigned i_blane = 0;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const float *)ldsf) + (unsigned)wave * 1024u;
  auto issue_begin = [&]() -> bool {
    const int iy = iy0 + i_ky, ix = ix0 + i_kx;
    const bool ok = gvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    i_blane = (ok ? (map_off + (unsigned)(iy * a.W + ix)) : map_off) * 32u + (unsigned)sh * 16u;
    return ok;
  };
  auto issue_pair = [&](int s, int i) {  // chunk i of both operands into stage buffer s
    glds16_s(i_wp, a_lane, lds0 + (unsigned)(s * STAGE + i * 1024) * 4u); i_wp += w_step;
    glds16_s(i_bp, i_blane, lds0 + (unsigned)(s * STAGE + OP_FLOATS + i * 1024) * 4u); i_bp += b_step;
  };
  auto issue_end = [&]() {
    if (++i_cg == spt) { i_cg = 0; i_bp = b_base; if (++i_kx == a.KW) { i_kx = 0; ++i_ky; } }
  };
  auto zero_oob = [&](int s, bool ok) {
    if (!ok) {
      float *B = ldsf + s * STAGE + OP_FLOATS + tid * 4;
#pragma unroll
      for (int i = 0; i < KCH; ++i) *reinterpret_cast<f32x4 *>(B + i * 1024) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int lane_off = l31 * 8 + half * 4;
  f32x4 af[2][2], bf[2][2];
  auto load_frags = [&](int s, int kk, int slot) {
    const float *Al = ldsf + s * STAGE + lane_off, *Bl = Al + OP_FLOATS;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) af[slot][mi] = *reinterpret_cast<const f32x4 *>(Al + (kk * 128 + wm * 64 + mi * 32) * 8);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[slot][ni] = *reinterpret_cast<const f32x4 *>(Bl + (kk * 128 + wn * 64 + ni * 32) * 8);
  };

  const int n_more = st1 - 1 - st0;  // stages that prefetch a successor
  const int b0 = n_more & 1;         // stage st lives in buffer (st - st0 + b0) & 1: the LAST stage is always in buffer 0
  bool ok_next = issue_begin();
#pragma unroll
  for (int i = 0; i < KCH; ++i) issue_pair(b0, i);
  issue_end();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  zero_oob(b0, ok_next);
  __syncthreads();
  load_frags(b0, 0, 0);

  auto body = [&](auto more_tag, auto parity_tag) {  // buffer parity is a compile-time tag: LDS offsets become immediates
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int s = decltype(parity_tag)::value;
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
      const int cur = kk & 1;
      if (kk == KCH - 1 && MORE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next stage this wave issued has landed
        zero_oob(s ^ 1, ok_next);
        __syncthreads();                                   // (+ lgkmcnt(0): this wave's last reads of stage s are done)
        load_frags(s ^ 1, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = t >> 2, mi = (t >> 1) & 1, ni = t & 1;
        __builtin_amdgcn_sched_barrier(0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi][j], bf[cur][ni][j], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0 && kk + 1 < KCH) load_frags(s, kk + 1, cur ^ 1);
        if constexpr (MORE) {
          if (kk == 0 && t == 0) ok_next = issue_begin();
          if (kk == 0 && t >= 1 && t <= KCH) issue_pair(s ^ 1, t - 1);
          if (kk == 0 && t == KCH) issue_end();
        }
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  {
    int st = st0;
    if (n_more & 1) { body(std::true_type{}, P1{}); ++st; }
    for (; st < st1 - 1; st += 2) { body(std::true_type{}, P0{}); body(std::true_type{}, P1{}); }
    body(std::false_type{}, P0{});
  }

  if (a.part) {  // raw partial sums
    float *slab = a.part + (size_t)blockIdx.z * (a.CoutP / 8) * a.pitch_out * 8;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const long long pix = p0 + wn * 64 + ni * 32 + l31;
      if (pix >= a.P) continue;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e];
          *reinterpret_cast<f32x4 *>(slab + ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4) = v;
        }
    }
    return;
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const long long pix = p0 + wn * 64 + ni * 32 + l31;
    if (pix >= a.P) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cout0 + wm * 64 + mi * 32) / 8 + g;
        if (cb >= a.Cb_out) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4);
        const size_t off = ((size_t)cb * a.pitch_out + (size_t)pix) * 8 + half * 4;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e] + b4[e];
        if (a.res) v += *reinterpret_cast<const f32x4 *>(a.res + off);
        if (a.relu && !(cb >= a.norelu_cb0 && cb < a.norelu_cb1)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.0f ? 0.0f : v[e];
        }
        *reinterpret_cast<f32x4 *>(a.out + off) = v;
      }
  }
}

// one thread = one 16-byte pixel record (8 channels)
__global__ void maxpool2d_c8i_bf16_kernel(const bf16_t *__restrict__ in, int Cb, int B, int H, int W, size_t pitch_in, int k, int stride, int pad,
                                          int OH, int OW, size_t pitch_out, bf16_t *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * B * OH * OW;
  if (t >= total) return;
  size_t r = t;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH); r /= OH;
  const int b = (int)(r % B); const size_t cb = r / B;
  float m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(in) + cb * pitch_in + (size_t)b * H * W;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) {
      const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const u32x4 v = ip[(size_t)iy * W + ix];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned w = v[e];
        const float lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xffff0000u);
        m[2 * e] = lo > m[2 * e] ? lo : m[2 * e];
        m[2 * e + 1] = hi > m[2 * e + 1] ? hi : m[2 * e + 1];
      }
    }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(m[2 * e], m[2 * e + 1]);
  reinterpret_cast<u32x4 *>(out)[cb * pitch_out + ((size_t)b * OH + oy) * OW + ox] = o;
}

// transformed image as bf16 C8I with TWO channel-block planes (channels 3..15 zero: the MFMA consumes chunk pairs)
__global__ void image_transform_c8i_bf16_kernel(const float *__restrict__ in, int H, int W, int s0, int s1, int s2, double scale, double m0,
                                                double m1, double m2, double d0, double d1, double d2, int has_std, bf16_t *__restrict__ out,
                                                size_t pitch) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t plane = (size_t)H * W;
  if (t >= plane) return;
  double v0 = (double)in[(size_t)s0 * plane + t], v1 = (double)in[(size_t)s1 * plane + t], v2 = (double)in[(size_t)s2 * plane + t];
  if (scale != 1.0) { v0 = v0 * scale; v1 = v1 * scale; v2 = v2 * scale; }
  v0 = v0 + (-m0); v1 = v1 + (-m1); v2 = v2 + (-m2);
  if (has_std) { v0 = v0 / d0; v1 = v1 / d1; v2 = v2 / d2; }
  u16x4 lo = {f2bf((float)v0), f2bf((float)v1), f2bf((float)v2), 0}, z = {0, 0, 0, 0};
  *reinterpret_cast<u16x4 *>(out + t * 8) = lo;
  *reinterpret_cast<u16x4 *>(out + t * 8 + 4) = z;
  *reinterpret_cast<u16x4 *>(out + (pitch + t) * 8) = z;
  *reinterpret_cast<u16x4 *>(out + (pitch + t) * 8 + 4) = z;
}

__global__ void roi_pool_c8i_bf16_kernel(const bf16_t *__restrict__ feat, int Cb, int H, int W, size_t pitch_f, const float *__restrict__ rois,
                                         int roi_stride, int N, int PH, int PW, float scale, bf16_t *__restrict__ out, size_t pitch_o, int roi_bins) {
  const int PP = PH * PW;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * Cb * PP * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int bin = (int)(r % PP); r /= PP;
  const int cb = (int)(r % Cb); const int n = (int)(r / Cb);
  const int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, RoiRule{1.0f, 0, roi_bins}, H, W, PH, PW, ph, pw, hs, he, ws, we);
  const bool empty = (he <= hs) || (we <= ws);
  f32x4 m = empty ? f32x4{0, 0, 0, 0} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const bf16_t *fp = feat + (size_t)cb * pitch_f * 8 + h * 4;
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {
      const u16x4 v = *reinterpret_cast<const u16x4 *>(fp + ((size_t)y * W + x) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float f = bf2f(v[e]); m[e] = f > m[e] ? f : m[e]; }
    }
  u16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(m[e]);
  *reinterpret_cast<u16x4 *>(out + ((size_t)cb * pitch_o + ((size_t)n * PH + ph) * PW + pw) * 8 + h * 4) = o;
}

// ---- ROI max-pooling, bf16, fast path ---------------------------------------------------------------------------------------
// The kernel above spends ~200 instructions per 8 output bytes (per-thread bin arithmetic, 4 channels per thread).  Here the
// feature map is first re-coded so that bf16 order is int16 order (negative values: magnitude bits flipped — a monotone,
// self-inverse map, so max commutes with it and the result is bit-identical), and a thread owns one (roi, bin) for a run of
// channel blocks: the bin arithmetic is done once, each pixel record (8 channels) costs one 16-byte load + 4 v_pk_max_i16, and a
// wave's stores are 64 consecutive records.
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf16x2_sortable(unsigned x) { return x ^ (((x >> 15) & 0x00010001u) * 0x7fffu); }

__global__ void bf16_sortable_kernel(const u32x4 *__restrict__ in, size_t n, u32x4 *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u32x4 v = in[t];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = bf16x2_sortable(v[e]);
  out[t] = v;
}

// vertical range-max (sparse) table level of the sortable map: out[y] = max(prev[y], prev[y + step]) = max over rows y .. y + 2 * step - 1
// (rows past H - 2 * step are never queried).  The fused ROI max-pool below then reads two rows per window column instead of all of them.
__global__ void vmax_level_sorted_kernel(const u32x4 *__restrict__ prev, u32x4 *__restrict__ out, int H, int W, size_t pitch, int Cb, int step) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (t >= HW * Cb) return;
  const int cb = (int)(t / HW); const size_t px = t - (size_t)cb * HW;
  const int y = (int)(px / W);
  const u32x4 a = prev[(size_t)cb * pitch + px];
  const u32x4 b = prev[(size_t)cb * pitch + px + (y + step < H ? (size_t)step * W : 0)];
  u32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned ua = a[e], ub = b[e];
    const i16x2 mx = __builtin_elementwise_max(__builtin_bit_cast(i16x2, ua), __builtin_bit_cast(i16x2, ub));
    r[e] = __builtin_bit_cast(unsigned, mx);
  }
  out[(size_t)cb * pitch + px] = r;
}

template <int CBG, bool NT = false>  // channel blocks per thread (grid.y = Cb / CBG); NT: non-temporal stores (timing experiment, debug flavour)
__global__ __launch_bounds__(256) void roi_pool_c8i_bf16_sorted_kernel(const u32x4 *__restrict__ feat, int H, int W, size_t pitch_f,
                                                                        const float *__restrict__ rois, int roi_stride, int N, int PH, int PW, float scale,
                                                                        u32x4 *__restrict__ out, size_t pitch_o, int roi_bins) {
  const int PP = PH * PW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (roi, bin): the output row of the pooled batch
  if (t >= (size_t)N * PP) return;
  const int n = (int)(t / PP), bin = (int)(t - (size_t)n * PP);
  const int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, RoiRule{1.0f, 0, roi_bins}, H, W, PH, PW, ph, pw, hs, he, ws, we);
  const bool empty = (he <= hs) || (we <= ws);
  const int cb0 = blockIdx.y * CBG;
  const u32x4 *fp = feat + (size_t)cb0 * pitch_f;
  u32x4 *op = out + (size_t)cb0 * pitch_o + t;
  const unsigned lowest = 0x80008000u;  // int16 minimum in both halves
  u32x4 m[CBG];
#pragma unroll
  for (int c = 0; c < CBG; ++c) m[c] = u32x4{lowest, lowest, lowest, lowest};
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {
      const size_t px = (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < CBG; ++c) {
        const u32x4 v = fp[(size_t)c * pitch_f + px];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned cur = m[c][e], val = v[e];  // scalar copies: __builtin_bit_cast of a vector-element lvalue reads element 0
          const i16x2 mx = __builtin_elementwise_max(__builtin_bit_cast(i16x2, cur), __builtin_bit_cast(i16x2, val));
          m[c][e] = __builtin_bit_cast(unsigned, mx);
        }
      }
    }
#pragma unroll
  for (int c = 0; c < CBG; ++c) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = empty ? 0u : bf16x2_sortable(m[c][e]);
    if constexpr (NT) __builtin_nontemporal_store(o, &op[(size_t)c * pitch_o]);
    else op[(size_t)c * pitch_o] = o;
  }
}

// max-pool OF the ROI-pooled map, straight from the feature map (graph heads: Inception's Mixed_7a pools its input — the ROI-pooled
// 17 x 17 x 768 tensor, 0.9 GB for 2000 ROIs — 3 x 3 / 2; max of maxes = one max over the union of the bins' windows).  Output pixel
// (oy, ox) covers bins [o * stride - pad, + k) clipped to the bin grid; consecutive bins touch or overlap (ceil((b + 1) h) >= floor((b + 1) h)),
// so the union of their windows is ONE window of the map; a bin that falls outside the map is empty and holds 0 in the pooled tensor, so 0
// joins the max when the group has one.  Bin bounds are the pooling kernel's own expressions; max is exact: the result is the two-step
// result bit for bit (up to the sign of a zero), and the max-pool launch that re-read the pooled tensor (210 us, 5.2 TB/s) is gone.
template <int CBG>
__global__ __launch_bounds__(256) void roi_maxpool_c8i_bf16_sorted_kernel(const u32x4 *__restrict__ feat, int H, int W, size_t pitch_f,
                                                                           const float *__restrict__ rois, int roi_stride, int N, int PH, int PW, float scale,
                                                                           int k, int stride, int pad, int OH, int OW, u32x4 *__restrict__ out, size_t pitch_o,
                                                                           const u32x4 *__restrict__ tabs, size_t level_elems, int n_levels) {
  const int OP = OH * OW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (roi, output pixel): the output row of the max-pooled batch
  if (t >= (size_t)N * OP) return;
  const int n = (int)(t / OP), o = (int)(t - (size_t)n * OP);
  const int oy = o / OW, ox = o - oy * OW;
  const float *ro = rois + (size_t)roi_stride * n;
  const int sw = (int)roundf((ro[1] - 1.0f) * scale), sh = (int)roundf((ro[2] - 1.0f) * scale);
  const int ew = (int)roundf((ro[3] - 1.0f) * scale), eh = (int)roundf((ro[4] - 1.0f) * scale);
  const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
  const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
  bool any_empty = false;
  int hs = 0x7fffffff, he = -1, ws = 0x7fffffff, we = -1;
  for (int ph = max(oy * stride - pad, 0); ph < min(oy * stride - pad + k, PH); ++ph) {
    int a = (int)floorf((float)ph * bh) + sh, b = (int)ceilf((float)(ph + 1) * bh) + sh;
    a = min(max(a, 0), H); b = min(max(b, 0), H);
    if (b <= a) any_empty = true; else { hs = min(hs, a); he = max(he, b); }
  }
  for (int pw = max(ox * stride - pad, 0); pw < min(ox * stride - pad + k, PW); ++pw) {
    int a = (int)floorf((float)pw * bw) + sw, b = (int)ceilf((float)(pw + 1) * bw) + sw;
    a = min(max(a, 0), W); b = min(max(b, 0), W);
    if (b <= a) any_empty = true; else { ws = min(ws, a); we = max(we, b); }
  }
  const int cb0 = blockIdx.y * CBG;
  const u32x4 *fp = feat + (size_t)cb0 * pitch_f;
  u32x4 *op = out + (size_t)cb0 * pitch_o + t;
  const unsigned lowest = 0x80008000u;  // int16 minimum in both halves
  u32x4 m[CBG];
#pragma unroll
  for (int c = 0; c < CBG; ++c) m[c] = u32x4{lowest, lowest, lowest, lowest};
  // rows [hs, he) as ceil(h / 2^lv) blocks of 2^lv rows from table level lv (the last block pulled back to end at he: overlap is
  // harmless for a max); lv = floor(log2 h) capped at the levels built -> two row reads per column for any window height
  int lv = 0, bstep = 1;
  if (tabs && he - hs > 1) {
    lv = min(31 - __builtin_clz((unsigned)(he - hs)), n_levels);
    bstep = 1 << lv;
    if (lv > 0) fp = tabs + (size_t)(lv - 1) * level_elems + (size_t)cb0 * pitch_f;
  }
  for (int y0 = hs; y0 < he; y0 += bstep) {
    const int y = min(y0, he - bstep);
    for (int x = ws; x < we; ++x) {
      const size_t px = (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < CBG; ++c) {
        const u32x4 v = fp[(size_t)c * pitch_f + px];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned cur = m[c][e], val = v[e];
          const i16x2 mx = __builtin_elementwise_max(__builtin_bit_cast(i16x2, cur), __builtin_bit_cast(i16x2, val));
          m[c][e] = __builtin_bit_cast(unsigned, mx);
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CBG; ++c) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned cur = m[c][e];
      if (any_empty) {  // an empty bin's 0.0 (sortable code 0) takes part in the max
        const i16x2 mx = __builtin_elementwise_max(__builtin_bit_cast(i16x2, cur), i16x2{0, 0});
        cur = __builtin_bit_cast(unsigned, mx);
      }
      r[e] = bf16x2_sortable(cur);
    }
    op[(size_t)c * pitch_o] = r;
  }
}

// average pool of the bf16 maps into the fp32 C8 matrix the (fp32) head GEMM reads.  Block = one channel block x 16 maps: the
// 16 * HW records are read as consecutive 16-byte loads (fully coalesced) into LDS as fp32, then 128 threads (map, channel) sum
// their HW values in pixel order (the order of the plain kernel below, so the result is bit-identical to it).
__global__ __launch_bounds__(256) void avgpool_c8i_bf16_to_c8_lds_kernel(const bf16_t *__restrict__ in, int N, int HW, size_t pitch, float inv,
                                                                          float *__restrict__ out, int Mp) {
  extern __shared__ float sm[];  // [16 * HW][8 + 1]: odd row stride -> the 128 summing threads hit distinct banks
  const int cb = blockIdx.y, n0 = blockIdx.x * 16;
  const int nm = min(16, N - n0);
  const int nrec = nm * HW;
  const u32x4 *src = reinterpret_cast<const u32x4 *>(in) + (size_t)cb * pitch + (size_t)n0 * HW;
  for (int i = threadIdx.x; i < nrec; i += 256) {
    const u32x4 v = src[i];
    float *d = sm + i * 9;
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[2 * e] = __uint_as_float(v[e] << 16); d[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  }
  __syncthreads();
  const int m = threadIdx.x >> 3, c = threadIdx.x & 7;
  if (threadIdx.x < 128 && m < nm) {
    const float *d = sm + (size_t)m * HW * 9 + c;
    float acc = 0.0f;
    for (int i = 0; i < HW; ++i) acc += d[i * 9];
    out[((size_t)cb * Mp + n0 + m) * 8 + c] = acc * inv;
  }
}


__global__ void avgpool_c8i_bf16_to_c8_kernel(const bf16_t *__restrict__ in, int N, int Cb, int HW, size_t pitch, float inv, float *__restrict__ out,
                                              int Mp) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * Cb * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int n = (int)(r % N); const int cb = (int)(r / N);
  const bf16_t *ip = in + ((size_t)cb * pitch + (size_t)n * HW) * 8 + h * 4;
  f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < HW; ++i) {
    const u16x4 v = *reinterpret_cast<const u16x4 *>(ip + (size_t)i * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) sacc[e] += bf2f(v[e]);
  }
  *reinterpret_cast<f32x4 *>(out + ((size_t)cb * Mp + n) * 8 + h * 4) = sacc * inv;
}

// nn.SpatialAveragePooling(kw,kh,sw,sh,pw,ph), count_include_pad (torch's default): sum over the in-map cells of the window
// in row-major order, divided by kh*kw regardless of how many cells lie in the padding.  T = float or bf16_t.
template <typename T>
__global__ void avgpool2d_c8i_kernel(const T *__restrict__ in, int Cb, int B, int H, int W, size_t pitch_in, int kh, int kw, int sh, int sw, int ph,
                                     int pw, int OH, int OW, size_t pitch_out, T *__restrict__ out, const float *__restrict__ bias, int relu) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * B * OH * OW * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH); r /= OH;
  const int b = (int)(r % B); const size_t cb = r / B;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) {
      const int iy = oy * sh + ky - ph, ix = ox * sw + kx - pw;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const size_t off = (cb * pitch_in + ((size_t)b * H + iy) * W + ix) * 8 + h * 4;
      if constexpr (sizeof(T) == 2) {
        const u16x4 v = *reinterpret_cast<const u16x4 *>(in + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += bf2f(v[e]);
      } else {
        acc += *reinterpret_cast<const f32x4 *>(in + off);
      }
    }
  const float inv = 1.0f / (float)(kh * kw);
  const size_t oo = (cb * pitch_out + ((size_t)b * OH + oy) * OW + ox) * 8 + h * 4;
  acc = acc * inv;
  if (bias) {  // a commuted pool -> pointwise-convolution pair (graph_parse): the convolution's bias and ReLU follow the pool
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] += bias[cb * 8 + h * 4 + e]; if (relu && acc[e] < 0.0f) acc[e] = 0.0f; }
  }
  if constexpr (sizeof(T) == 2) {
    u16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[e]);
    *reinterpret_cast<u16x4 *>(out + oo) = o;
  } else {
    *reinterpret_cast<f32x4 *>(out + oo) = acc;
  }
}

// bf16: one thread = one 16-byte pixel record; same summation order per channel as the template above
__global__ void avgpool2d_c8i_bf16_kernel(const bf16_t *__restrict__ in, int Cb, int B, int H, int W, size_t pitch_in, int kh, int kw, int sh, int sw,
                                          int ph, int pw, int OH, int OW, size_t pitch_out, bf16_t *__restrict__ out, const float *__restrict__ bias, int relu) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * B * OH * OW;
  if (t >= total) return;
  size_t r = t;
  const int ox = (int)(r % OW); r /= OW;
  const int oy = (int)(r % OH); r /= OH;
  const int b = (int)(r % B); const size_t cb = r / B;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(in) + cb * pitch_in + (size_t)b * H * W;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) {
      const int iy = oy * sh + ky - ph, ix = ox * sw + kx - pw;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const u32x4 v = ip[(size_t)iy * W + ix];
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(v[e] << 16); acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u); }
    }
  const float inv = 1.0f / (float)(kh * kw);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] *= inv;
    if (bias) { acc[e] += bias[cb * 8 + e]; if (relu && acc[e] < 0.0f) acc[e] = 0.0f; }
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
  reinterpret_cast<u32x4 *>(out)[cb * pitch_out + ((size_t)b * OH + oy) * OW + ox] = o;
}

// stride-1 average pooling of SMALL maps (per-ROI 8x8 in Inception's Mixed_7b/7c: H*W <= 256) through LDS: a block owns
// 256 / (H*W) consecutive maps of one channel block, reads their records once (consecutive 16-byte loads) and takes the window
// cells from LDS — the plain kernel issues kh*kw global loads per output.  Same cells, same order: bit-identical to it.
__global__ __launch_bounds__(256) void avgpool2d_c8i_bf16_small_kernel(const bf16_t *__restrict__ in, int B, int H, int W, size_t pitch_in, int kh, int kw,
                                                                        int ph, int pw, size_t pitch_out, bf16_t *__restrict__ out,
                                                                        const float *__restrict__ bias, int relu) {
  __shared__ u32x4 tile[256];
  const int HW = H * W, mpb = 256 / HW;           // maps per block
  const int cb = blockIdx.y;
  const int b0 = blockIdx.x * mpb;
  const int nrec = min(mpb, B - b0) * HW;
  const int t = threadIdx.x;
  const u32x4 *ip = reinterpret_cast<const u32x4 *>(in) + (size_t)cb * pitch_in + (size_t)b0 * HW;
  if (t < nrec) tile[t] = ip[t];
  __syncthreads();
  if (t >= nrec) return;
  const int m = t / HW, r = t - m * HW;
  const int oy = r / W, ox = r - oy * W;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
  for (int ky = 0; ky < kh; ++ky)
    for (int kx = 0; kx < kw; ++kx) {
      const int iy = oy + ky - ph, ix = ox + kx - pw;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const u32x4 v = tile[m * HW + iy * W + ix];
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(v[e] << 16); acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u); }
    }
  const float inv = 1.0f / (float)(kh * kw);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] *= inv;
    if (bias) { acc[e] += bias[cb * 8 + e]; if (relu && acc[e] < 0.0f) acc[e] = 0.0f; }
  }
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
  reinterpret_cast<u32x4 *>(out)[(size_t)cb * pitch_out + (size_t)b0 * HW + t] = o;
}

// ------------------------------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------------------------------
// nn.SpatialCrossMapLRN(size, alpha, beta, k) on C8I (models/alexnet.lua's trunk: norm1 / norm2): one thread per (pixel row, channel
// block); the window spans at most one block either side (size <= 17).  fp32, squares summed in ascending channel order.
__global__ void lrn_c8i_kernel(const float *__restrict__ in, int Cb, int C, size_t rows, size_t pitch_in, int size, float alpha, float beta, float k,
                               size_t pitch_out, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * Cb) return;
  const size_t row = t % rows;
  const int cb = (int)(t / rows);
  float v[24];  // channels 8*cb - 8 .. 8*cb + 15
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int b = cb - 1 + q;
    f32x4 lo = f32x4{0, 0, 0, 0}, hi = lo;
    if (b >= 0 && b < Cb) {
      const float *p = in + ((size_t)b * pitch_in + row) * 8;
      lo = *reinterpret_cast<const f32x4 *>(p); hi = *reinterpret_cast<const f32x4 *>(p + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[q * 8 + e] = lo[e]; v[q * 8 + 4 + e] = hi[e]; }
  }
  const int half = (size - 1) / 2;
  const float a = alpha / (float)size;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cb * 8 + j;
    float ssum = 0.0f;
#pragma unroll
    for (int d = -8; d <= 8; ++d) {  // static register indices; the window is [-half, half]
      const int cc = c + d;
      if (d >= -half && d <= half && cc >= 0 && cc < C) { const float x = v[8 + j + d]; ssum = ssum + x * x; }
    }
    const float sc = k + a * ssum;
    o[j] = c < C ? v[8 + j] * powf(sc, -beta) : 0.0f;
  }
  float *q = out + ((size_t)cb * pitch_out + row) * 8;
  *reinterpret_cast<f32x4 *>(q) = f32x4{o[0], o[1], o[2], o[3]};
  *reinterpret_cast<f32x4 *>(q + 4) = f32x4{o[4], o[5], o[6], o[7]};
}

// C8I map (B = 1) <-> C8P map (dense.h: one zero halo row / column in front, padded behind): the trunk's 3x3 / stride-1 / pad-1
// convolutions run on dense.hip's Winograd F(2x2,3x3) kernel, which reads its halo tiles straight from the padded layout
__global__ void c8i_to_c8p_kernel(const float *__restrict__ in, int Cb, int H, int W, size_t pitch, int Hp, int Wp, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (t >= HW * Cb * 2) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const size_t px = r % HW; const int cb = (int)(r / HW);
  const int y = (int)(px / W), x = (int)(px - (size_t)y * W);
  *reinterpret_cast<f32x4 *>(out + (((size_t)cb * Hp + y + 1) * Wp + x + 1) * 8 + h * 4) = *reinterpret_cast<const f32x4 *>(in + ((size_t)cb * pitch + px) * 8 + h * 4);
}
__global__ void c8p_to_c8i_kernel(const float *__restrict__ in, int Cb, int H, int W, int Hp, int Wp, size_t pitch, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t HW = (size_t)H * W;
  if (t >= HW * Cb * 2) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const size_t px = r % HW; const int cb = (int)(r / HW);
  const int y = (int)(px / W), x = (int)(px - (size_t)y * W);
  *reinterpret_cast<f32x4 *>(out + ((size_t)cb * pitch + px) * 8 + h * 4) = *reinterpret_cast<const f32x4 *>(in + (((size_t)cb * Hp + y + 1) * Wp + x + 1) * 8 + h * 4);
}

// one-map C8I -> pixel-major [y][x][Cb * 8] (dense.h roi_pool_pm's operand: a pixel's channels contiguous)
__global__ void c8i_to_pixel_major_kernel(const float *__restrict__ in, int Cb, size_t HW, size_t pitch, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= HW * Cb * 2) return;
  const int q = (int)(t % (Cb * 2)); const size_t px = t / (Cb * 2);   // q = cb * 2 + half: consecutive lanes write consecutive 16 bytes
  *reinterpret_cast<f32x4 *>(out + px * (size_t)Cb * 8 + q * 4) = *reinterpret_cast<const f32x4 *>(in + ((size_t)(q >> 1) * pitch + px) * 8 + (q & 1) * 4);
}

// A batch of small per-ROI maps (ResNet's layer4: 1000 x 7 x 7) as ONE padded C8P image for the Winograd kernel: a mosaic of
// (H + 1) x (W + 1) cells, mx maps per mosaic row, map b at cell (b / mx, b % mx) with its pixels in the cell's top-left H x W and
// the cell's last row / column left ZERO — together with the C8P halo every map is surrounded by zeros, i.e. pad 1.  The
// converters only ever touch map pixels, so the zeros written at allocation stay (any N up to the capacity the mosaic was laid out for).
__global__ void c8i_to_mosaic_kernel(const float *__restrict__ in, int Cb, int N, int H, int W, size_t pitch_i, int mx, int Hp, int Wp, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rows = (size_t)N * H * W;
  if (t >= rows * Cb * 2) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const size_t row = r % rows; const int cb = (int)(r / rows);
  const int b = (int)(row / (H * W)), px = (int)(row - (size_t)b * H * W), y = px / W, x = px - y * W;
  const size_t my = (size_t)(b / mx) * (H + 1) + y + 1, mxx = (size_t)(b % mx) * (W + 1) + x + 1;
  *reinterpret_cast<f32x4 *>(out + (((size_t)cb * Hp + my) * Wp + mxx) * 8 + h * 4) = *reinterpret_cast<const f32x4 *>(in + ((size_t)cb * pitch_i + row) * 8 + h * 4);
}
__global__ void mosaic_to_c8i_kernel(const float *__restrict__ in, int Cb, int N, int H, int W, int mx, int Hp, int Wp, size_t pitch_o, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rows = (size_t)N * H * W;
  if (t >= rows * Cb * 2) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const size_t row = r % rows; const int cb = (int)(r / rows);
  const int b = (int)(row / (H * W)), px = (int)(row - (size_t)b * H * W), y = px / W, x = px - y * W;
  const size_t my = (size_t)(b / mx) * (H + 1) + y + 1, mxx = (size_t)(b % mx) * (W + 1) + x + 1;
  *reinterpret_cast<f32x4 *>(out + ((size_t)cb * pitch_o + row) * 8 + h * 4) = *reinterpret_cast<const f32x4 *>(in + (((size_t)cb * Hp + my) * Wp + mxx) * 8 + h * 4);
}

// First layer (Cin = 3, fp32 graphs): im2col rows for the tuned GEMM.  One 8-channel record of the C8I image holds 3 real channels,
// so the tap-by-tap convolution spends 8 / 3 of the matrix work on zeros (and runs the 8-channel-stage kernel); here
// k = (ky * KW + kx) * 3 + c, zero beyond KH * KW * 3 and outside the image, written as the C8 matrix [K64 / 8][pitch][8].  It reads
// the transformed image's [3][H][W] planes: a wave's 64 pixels are then 64 * stride floats of one row, a few cache lines per load
// (from the C8I records each lane touched its own line: 59 us for AlexNet's conv1 instead of ~25).
__global__ void im2col3_c8i_kernel(const float *__restrict__ img, int H, int W, int KH, int KW, int sh, int sw, int ph, int pw, int OH, int OW, int nkb,
                                   size_t pitch_o, float *__restrict__ out) {
  const size_t P = (size_t)OH * OW;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * nkb) return;
  const int kb = (int)(t / P);
  const size_t pix = t - (size_t)kb * P;
  const int oy = (int)(pix / OW), ox = (int)(pix - (size_t)oy * OW);
  const int iy0 = oy * sh - ph, ix0 = ox * sw - pw;
  // (ky, kx, c) of k = kb * 8 once, then stepped: no division per element
  int tap = (kb * 8) / 3, c = kb * 8 - tap * 3;
  int ky = tap / KW, kx = tap - ky * KW;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int iy = iy0 + ky, ix = ix0 + kx;
    const bool ok = ky < KH && iy >= 0 && iy < H && ix >= 0 && ix < W;
    v[j] = ok ? img[((size_t)c * H + iy) * W + ix] : 0.0f;
    if (++c == 3) { c = 0; if (++kx == KW) { kx = 0; ++ky; } }
  }
  float *q = out + ((size_t)kb * pitch_o + pix) * 8;
  *reinterpret_cast<f32x4 *>(q) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4 *>(q + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// [Cout][3][KK] -> [Cout][KK * 3] (tap-major, the im2col k order)
__global__ void permute_w3_kernel(const float *__restrict__ w, int Cout, int KK, float *__restrict__ o) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Cout * KK * 3) return;
  const int n = t / (KK * 3), k = t - n * KK * 3, tap = k / 3, c = k - tap * 3;
  o[t] = w[((size_t)n * 3 + c) * KK + tap];
}

struct RnConv {
  int Cin = 0, Cout = 0, K = 0, stride = 1, pad = 0;  // square form (ResNet); KH/KW/sh/sw/ph/pw below are what the kernels use
  int KH = 0, KW = 0, sh = 1, sw = 1, ph = 0, pw = 0;
  float *wpk = nullptr, *bpk = nullptr;
  float *lin_w = nullptr, *lin_b = nullptr;  // 1x1 / stride 1 with Cin % 64 == 0: also packed for the tuned GEMM (linear_c8)
  float *col_w = nullptr, *col_b = nullptr;  // Cin == 3 (fp32): packed for the GEMM over im2col rows, k = tap * 3 + c
  float *wino = nullptr;                     // graph trunks / ResNet heads (fp32), 3x3 / stride 1 / pad 1: Winograd-transformed weights (dense.h conv3x3_c8p)
  float *mos_in = nullptr, *mos_out = nullptr;  // ResNet heads: the mosaic images (shared by the head's eligible convolutions), laid out for
  int mos_h = 0, mos_w = 0, mos_mx = 0, mos_rows = 0, mos_cols = 0;  //   mos_rows x mos_cols px = max_rois maps of mos_h x mos_w, mos_mx per mosaic row
  bf16_t *wpk16 = nullptr;                   // bf16 graph: [tap][nch2][CoutP][8]
  int norelu_c0 = 0, norelu_c1 = 0;          // output channels [c0, c1) skip the ReLU (fused siblings with mixed activations; multiples of 8)
  float *ws = nullptr;                       // the graph's split-K workspace
  size_t ws_bytes = 0;
};
struct RnBlock {
  std::vector<RnConv> convs;
  bool has_sc = false;
  RnConv sc;
};
struct GTensor {
  int C = 0, H = 0, W = 0;
  float *buf = nullptr;
  float *c8p = nullptr;       // trunk tensors a Winograd convolution reads or writes: the same map in the C8P layout
  int p_h = 0, p_w = 0;       // the dims c8p's zero halo is laid out for (re-zeroed when the image size changes)
  int alias_of = -1, alias_c_off = 0;  // >= 0: this tensor is channels [alias_c_off, alias_c_off + C) of tensor alias_of (fused sibling convolutions)
};
struct GOp {
  int kind = 0, src = 0, dst = 0, dst_c_off = 0, kh = 1, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0, relu = 0;
  int src_c_off = 0, cin = 0, ceil_mode = 0;   // the op reads channels [src_c_off, src_c_off + cin) of src
  float lrn_alpha = 0.f, lrn_beta = 0.f, lrn_k = 1.f;
  RnConv conv;
  float *pool_bias = nullptr;  // average pool only: bias (+ ReLU if relu) applied after the pool (commuted pool -> pointwise convolution)
  bool from_rois = false;      // head graphs: a max-pool of the ROI-pooled input itself -> computed from the feature map (roi_maxpool_c8i_bf16_sorted_kernel)
  float *fc_w = nullptr, *fc_b = nullptr;  // head graphs (fp32): a convolution over the WHOLE pooled map (AlexNet's fc6) packed for the tuned GEMM, K = (channel block, bin)
};
struct ResNetGraph {
  // op-list mode (graph_build): branching graphs; tensor 0 = image (trunk) / ROI-pooled map (head)
  bool is_graph = false;
  std::vector<GOp> g_trunk;
  std::vector<std::vector<GOp>> g_heads;  // one op list per tower (1 for plain inceptionv3.lua), all over the same head tensors
  std::vector<GTensor> t_trunk, t_head;
  int feat_tensor = 0, out_tensor = 0;
  RnConv conv1;
  std::vector<RnBlock> trunk;
  std::vector<std::vector<RnBlock>> heads;  // one layer4 copy per tower (1 for plain resnet.lua)
  int feat_c = 0, out_c = 0, pooled = 14, max_rois = 0;
  bool bf16 = false;             // activations / conv weights in bf16 (fp32 accumulate); the cls / bbox head GEMM stays fp32
  float *img = nullptr;          // C8I image
  float *img_planar = nullptr;   // fp32 graphs: the transformed image as [3][H][W] planes too (im2col first layer)
  float *tb[4] = {nullptr, nullptr, nullptr, nullptr};  // trunk activations (rotating)
  float *hb[4] = {nullptr, nullptr, nullptr, nullptr};  // per-ROI head activations (rotating)
  size_t tb_elems = 0, hb_elems = 0;
  float *feat = nullptr;         // points into tb[]: layer3 output of the last trunk run
  int feat_h = 0, feat_w = 0, last_h = -1, last_w = -1;
  float *fc_x = nullptr;         // (bin, roi)-row ROI-pooled matrix [Cb][PP][round_up(max_rois, 128)][8] for a GOp with fc_w
  bf16_t *feat_sorted = nullptr; // order-preserving int16 re-coding of the cached feature map (bf16 ROI pooling), rebuilt per trunk run
  size_t feat_sorted_elems = 0;
  bool feat_sorted_valid = false;
  bf16_t *feat_vmax = nullptr;   // vertical range-max levels 1 .. feat_vmax_levels of feat_sorted (the fused ROI max-pool reads two rows per column)
  size_t feat_vmax_elems = 0;    // elements per level
  int feat_vmax_levels = 0;
  bool feat_vmax_valid = false;
  float *splitk_ws = nullptr;    // fp32 partial slabs of split-K convolutions (bf16 graph; one stream at a time, like tb / hb)
  int roi_bins = 0;              // MPN_ROI_BINS_* (resnet_set_roi_bins)
  // second tower LANE (graphs with more than one head): a second set of per-ROI activation buffers, so that towers 1, 3 can run on a
  // second stream beside towers 0, 2, 4 (resnet_head_forward's `lane`; pipeline.hip run_detect).  Weights, the feature map and its sorted /
  // range-max images are shared (read-only while the towers run: resnet_heads_prepare builds them before the lanes fork).
  std::vector<GTensor> t_head2;
  float *hb2[4] = {nullptr, nullptr, nullptr, nullptr};
  bool has_lane2 = false;
  std::vector<void *> allocs;
};
constexpr size_t SPLITK_WS_BYTES = (size_t)96 << 20;

static int rn_alloc(ResNetGraph *g, float **p, size_t bytes) {
  void *q = nullptr;
  MPN_CHECK_HIP(hipMalloc(&q, bytes ? bytes : 4));
  g->allocs.push_back(q);
  *p = static_cast<float *>(q);
  return MPN_OK;
}

// a buffer that grows with the image size (outside the steady state): the old block is released (hipFree waits for the device, so
// kernels still reading it have finished) instead of staying on g->allocs until the handle dies
static int rn_regrow(ResNetGraph *g, float **p, size_t bytes) {
  if (*p) {
    auto it = std::find(g->allocs.begin(), g->allocs.end(), static_cast<void *>(*p));
    if (it != g->allocs.end()) g->allocs.erase(it);
    MPN_CHECK_HIP(hipFree(*p));
    *p = nullptr;
    bump_alloc_generation();  // captured launch graphs (pipeline.hip) hold the old pointer
  }
  return rn_alloc(g, p, bytes);
}

MPN_KNOB(int, g_graph_fuse, 511);  // mpn_debug_set_graph_fuse: bit 0 = fuse sibling pointwise convolutions, bit 1 = commute average-pool -> pointwise convolution, bit 2 = max-pools of the ROI-pooled input computed from the feature map, bit 3 = fully-connected head layers (whole-map / 1x1-map convolutions) on the tuned GEMM, bit 4 = the image layer (Cin = 3) as a GEMM over im2col rows, bit 5 = trunk 3x3 / stride-1 convolutions on dense.hip's Winograd kernel, bit 6 = the fused ROI max-pool (bit 2) reads vertical range-max tables of the map, bit 7 = ResNet heads' 3x3 / stride-1 convolutions on the Winograd kernel (mosaic image of the per-ROI maps), bit 8 = the fully-connected operand pooled by dense.hip's pixel-major kernel; 0 = run the op list as given
static int rn_pack(ResNetGraph *g, RnConv &c, const float *d_w, const float *d_b) {
  if (g->bf16) {
    const int nch2 = round_up((c.Cin + 7) / 8, 2), CoutP = round_up(c.Cout, 128), KK = c.KH * c.KW;
    const size_t total = (size_t)KK * nch2 * CoutP * 8;
    float *w16 = nullptr;
    int rc = rn_alloc(g, &w16, total * sizeof(bf16_t));
    if (rc) return rc;
    c.wpk16 = reinterpret_cast<bf16_t *>(w16);
    if (!g->splitk_ws) {
      rc = rn_alloc(g, &g->splitk_ws, SPLITK_WS_BYTES);
      if (rc) return rc;
    }
    c.ws = g->splitk_ws; c.ws_bytes = SPLITK_WS_BYTES;
    rc = rn_alloc(g, &c.bpk, (size_t)CoutP * sizeof(float));
    if (rc) return rc;
    MPN_CHECK_HIP(hipMemset(c.bpk, 0, (size_t)CoutP * sizeof(float)));
    if (d_b) MPN_CHECK_HIP(hipMemcpy(c.bpk, d_b, (size_t)c.Cout * sizeof(float), hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, nullptr, d_w, c.Cin, c.Cout, KK, nch2, CoutP, c.wpk16);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  const int nch = (c.Cin + 7) / 8, CoutP = round_up(c.Cout, 128), KK = c.KH * c.KW;
  const size_t total = (size_t)KK * nch * CoutP * 8;
  int rc = rn_alloc(g, &c.wpk, total * sizeof(float));
  if (rc) return rc;
  if (!g->splitk_ws) {
    rc = rn_alloc(g, &g->splitk_ws, SPLITK_WS_BYTES);
    if (rc) return rc;
  }
  c.ws = g->splitk_ws; c.ws_bytes = SPLITK_WS_BYTES;
  rc = rn_alloc(g, &c.bpk, (size_t)CoutP * sizeof(float));
  if (rc) return rc;
  const size_t threads = total > (size_t)CoutP ? total : (size_t)CoutP;
  hipLaunchKernelGGL(pack_conv_generic_kernel, dim3((unsigned)cdiv_sz(threads, 256)), dim3(256), 0, nullptr, d_w, d_b, c.Cin, c.Cout, KK, nch, CoutP,
                     c.wpk, c.bpk);
  MPN_CHECK_LAUNCH();
  if (c.Cin == 3 && KK > 1 && (g_graph_fuse & 16)) {  // the image layer: GEMM over im2col rows
    const int Kc = KK * 3;
    float *perm = nullptr;
    rc = rn_alloc(g, &perm, (size_t)c.Cout * Kc * sizeof(float));
    if (rc == MPN_OK) rc = rn_alloc(g, &c.col_w, lin_wpk_elems(round_up(Kc, 64), c.Cout) * sizeof(float));
    if (rc == MPN_OK) rc = rn_alloc(g, &c.col_b, (size_t)lin_np(c.Cout) * sizeof(float));
    if (rc) return rc;
    hipLaunchKernelGGL(permute_w3_kernel, dim3((unsigned)cdiv(c.Cout * Kc, 256)), dim3(256), 0, nullptr, d_w, c.Cout, KK, perm);
    MPN_CHECK_LAUNCH();
    rc = pack_linear_weights(perm, d_b, Kc, c.Cout, 1, c.col_w, c.col_b, nullptr);
    if (rc) return rc;
  }
  if (c.KH == 1 && c.KW == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0 && c.Cin % 64 == 0) {  // a pointwise convolution IS a GEMM over the C8I rows
    rc = rn_alloc(g, &c.lin_w, lin_wpk_elems(c.Cin, c.Cout) * sizeof(float));
    if (rc) return rc;
    rc = rn_alloc(g, &c.lin_b, (size_t)lin_np(c.Cout) * sizeof(float));
    if (rc) return rc;
    rc = pack_linear_weights(d_w, d_b, c.Cin, c.Cout, 1, c.lin_w, c.lin_b, nullptr);
    if (rc) return rc;
  }
  return MPN_OK;
}

MPN_KNOB(unsigned long long *, g_bf16_trace, nullptr);  // mpn_debug_set_bf16_trace (tools/dma_trace.py)
MPN_KNOB(int, g_bf16_trace_kh, 3);
MPN_KNOB(int, g_bf16_fast_pool, 3);       // bit 0: row-per-thread ROI pooling (bf16: on the int16-sortable map), bit 1: LDS average pooling (bf16) (0 = the plain kernels; mpn_debug_set_bf16_fast_pool)
MPN_KNOB(int, g_fp32_pf, 1);              // fp32 graph: conv2d_c8i_pf_kernel for 32-channel-stage layers (0 = conv2d_c8i_kernel<4>; mpn_debug_set_fp32_pf)
MPN_KNOB(int, g_split_max_tiles, 192);     // split-K only layers with fewer 128 x 128 tiles than this (mpn_debug_set_split_max_tiles)
MPN_KNOB(int, g_bf16_split_target, 256);  // split-K (both dtypes) aims at this many blocks (0 = never split; mpn_debug_set_bf16_split_target)
MPN_KNOB(int, g_bf16_dma_tn, 0);  // 0 = pick per layer; 128 / 256 = force 256 couts x that many pixels, 1256 = 128 couts x 256 pixels (mpn_debug_set_bf16_dma_tn)
MPN_KNOB(int, g_bf16_dma, 1);  // mpn_debug_set_bf16_dma: 0 = never, 1 = large layers, 2 = every eligible layer (tests)
MPN_KNOB(int, g_bf16_exp, 0);   // mpn_debug_set_bf16_exp: GConvArgsB::exp (scheduling experiments of the LDS-DMA kernel)
#ifdef MPN_DEBUG_HOOKS
MPN_KNOB(int, g_bf16_nch, 4);   // mpn_debug_set_bf16_nch: 8 = 64-channel stages where Cin % 64 == 0 (one block per CU; measured slower, debug flavour only)
#endif
MPN_KNOB(int, g_bf16_bdir, 1);  // mpn_debug_set_bf16_bdir: 1 = conv2d_c8i_bf16_bdir_kernel for the large layers it measured faster on (all but strided pointwise ones), 0 = never (the LDS-DMA kernel), 2 = every eligible large layer
#ifdef MPN_DEBUG_HOOKS
MPN_KNOB(int, g_bf16_bdir_abl, 0);  // mpn_debug_set_bf16_bdir_abl: the B-direct kernel's ABL knock-outs (only in -DMPN_BF16_ABLATE builds)
MPN_KNOB(int, g_bf16_bdir_ver, 1);  // mpn_debug_set_bf16_bdir_ver: 1 = compiler-counted form <3>, 2 / 3 = hand-counted form <3> / <4> (debug flavour only)
#endif
#ifdef MPN_DEBUG_HOOKS
// mpn_debug_set_tower_knock (tools/tower_knockout.py; timing only, garbage results): the CEILING of a first per-ROI layer that never reads a
// materialised pooled tensor (VERDICT r5 task 1, step 1).  bit 0: the ROI pooling launch of the bf16 heads is skipped; bit 1: the convolutions
// that read the pooled tensor (Mixed_7a's fused 1x1 768 -> 384; layer4 block 1's conv1 and shortcut) run the B-direct kernel with every pixel
// load hitting ONE resident 1-KiB window (ABL bit 5: nothing can fetch its operand cheaper than from the CU's own cache).
static int g_tower_knock = 0;
static int g_pool_exp = 0;   // mpn_debug_set_pool_exp: timing experiments of the bf16 ROI pooling launch (resnet_head_forward)
static int g_knock_arm = 0;  // the next `g_knock_arm` rn_conv calls are first-layer convolutions (armed by the caller)
#endif
MPN_KNOB(int, g_roi_invariant, 1);  // mpn_debug_set_roi_invariant: 0 = per-ROI layers pick kernel / split by batch size as round 3 did (tests, timing)
// per_roi: the batch axis counts ROIs (the head of a graph model).  A ROI's result must not depend on which other ROIs share the
// launch (memoryEfficientForward's chunked == full, ImageDetect.lua:126-133; the ROI-sharded mode == the unsharded one), so for
// these layers everything that changes the summation ORDER is a function of the layer alone, never of in.B: no split-K (every
// kernel family below accumulates K = (tap, channel chunk) in ONE chain per output, and the bf16 tile shapes / the small and the
// LDS-DMA kernel share that chain bit for bit), pointwise convolutions always on the un-split GEMM, eligible 3x3 convolutions
// always on the Winograd mosaic with a batch-independent launch plan.  Tile shapes may still follow the batch size.
static int rn_conv(const RnConv &c, ActI in, float *out, const float *res, int relu, hipStream_t s, ActI *o, bool allow_gemm = true, bool per_roi = false) {
  const bool inv = per_roi && g_roi_invariant;
  if (c.wpk16) {  // bf16 graph
    GConvArgsB b{};
    b.in = reinterpret_cast<const bf16_t *>(in.p); b.wpk = c.wpk16; b.res = reinterpret_cast<const bf16_t *>(res); b.bpk = c.bpk;
    b.out = reinterpret_cast<bf16_t *>(out);
    b.B = in.B; b.H = in.H; b.W = in.W; b.nch2 = round_up((c.Cin + 7) / 8, 2);
    b.CoutP = round_up(c.Cout, 128); b.Cb_out = (c.Cout + 7) / 8;
    b.KH = c.KH; b.KW = c.KW; b.sh = c.sh; b.sw = c.sw; b.ph = c.ph; b.pw = c.pw;
    b.OH = (in.H + 2 * c.ph - c.KH) / c.sh + 1; b.OW = (in.W + 2 * c.pw - c.KW) / c.sw + 1;
    b.relu = relu; b.norelu_cb0 = c.norelu_c0 / 8; b.norelu_cb1 = c.norelu_c1 / 8;
    MPN_CHECK_ARG(in.C == c.Cin && b.OH > 0 && b.OW > 0);
    b.P = (long long)in.B * b.OH * b.OW;
    *o = ActI{out, in.B, c.Cout, b.OH, b.OW};
    b.pitch_in = in.pitch(); b.pitch_out = o->pitch();
    // round 4: the B-direct kernel (128 couts x 256 pixels; pixels straight into registers), where both tensors fit a 2-GiB descriptor
    // Where it measured faster, layer by layer against the LDS-DMA kernel on the per-ROI layers of configs[3] / [4] (tools/bench_conv_bf16.py,
    // profiles/r04_bf16_conv_bdirect.txt): convolutions with spatial taps at stride 1 (every pixel record is re-read once per tap, from
    // the CU's own cache: 1x3 / 3x1 -17 %, 3x3 -21 %, 1x7 / 7x1 -10 %), strided 3x3 with >= 512 input channels (-4 %), and pointwise
    // layers with Cin >= 2 Cout (2048 -> 512: -9 %).  NOT on pointwise layers with many output channels (512 -> 2048 + residual: +9 %:
    // its 128-cout tile re-reads the pixel tile twice as often as the 256-cout LDS-DMA shape), nor on strided pointwise layers
    // (every second record of a row: half-used cache lines per fragment load, +26 %).  A rule of the layer alone, never of the batch.
    const bool pointwise = b.KH == 1 && b.KW == 1, strided = b.sh > 1 || b.sw > 1;
    const bool bdir_wins = pointwise ? (!strided && c.Cin >= 2 * c.Cout) : (!strided || c.Cin >= 512);
#ifdef MPN_DEBUG_HOOKS
    const bool knock_first = g_knock_arm > 0 && (g_tower_knock & 2);
#else
    constexpr bool knock_first = false;
#endif
    if (g_bf16_bdir && (g_bf16_bdir == 2 || bdir_wins || knock_first) && g_bf16_dma && b.nch2 % 4 == 0 && b.P >= (g_bf16_dma == 2 ? 1 : 256 * 128) &&
        (size_t)b.nch2 * b.pitch_in * 16 < ((size_t)1 << 31) && (size_t)b.KH * b.KW * b.nch2 * b.CoutP * 16 < ((size_t)1 << 31)) {
      const int nx = (int)((b.P + 255) / 256), ny = b.CoutP / 128;
      const dim3 gridd((unsigned)(((nx + 7) / 8) * 8 * ny));
#ifdef MPN_DEBUG_HOOKS
      if (g_knock_arm > 0 && (g_tower_knock & 2)) {
        --g_knock_arm;
        hipLaunchKernelGGL((conv2d_c8i_bf16_bdir_kernel<3, 32 | 512>), gridd, dim3(256), 0, s, b, nx, ny);
        MPN_CHECK_LAUNCH();
        return MPN_OK;
      }
      if (g_bf16_bdir_ver == 9) {
        const int nxq = (nx + 7) / 8;
        const dim3 gridp((unsigned)(8 * ((nxq * ny + 1) / 2)));
        hipLaunchKernelGGL((conv2d_c8i_bf16_bdpp_kernel<3>), gridp, dim3(512), 0, s, b, nx, ny);
        MPN_CHECK_LAUNCH();
        return MPN_OK;
      }
      if (g_bf16_bdir_ver == 8) {
        const int ny8 = (b.CoutP + 255) / 256;
        const dim3 grid8((unsigned)(((nx + 7) / 8) * 8 * ny8));
#ifdef MPN_BF16_ABLATE
#define MPN_BDIR8_ABL(v) if (g_bf16_bdir_abl == v) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir8_kernel<4, v>), grid8, dim3(512), 0, s, b, nx, ny8); else
        MPN_BDIR8_ABL(1) MPN_BDIR8_ABL(2) MPN_BDIR8_ABL(8) MPN_BDIR8_ABL(16) MPN_BDIR8_ABL(32) MPN_BDIR8_ABL(7) MPN_BDIR8_ABL(23) MPN_BDIR8_ABL(24)
#undef MPN_BDIR8_ABL
#endif
        if (g_bf16_bdir_abl == 1000) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir8_kernel<8>), grid8, dim3(512), 0, s, b, nx, ny8);
        else if (g_bf16_bdir_abl == 1001) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir8_kernel<6>), grid8, dim3(512), 0, s, b, nx, ny8);
        else
        hipLaunchKernelGGL((conv2d_c8i_bf16_bdir8_kernel<4>), grid8, dim3(512), 0, s, b, nx, ny8);
        MPN_CHECK_LAUNCH();
        return MPN_OK;
      }
#endif
#ifdef MPN_DEBUG_HOOKS
      if (g_bf16_bdir_ver == 2) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir2_kernel<3>), gridd, dim3(256), 0, s, b, nx, ny);
      else if (g_bf16_bdir_ver == 3) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir2_kernel<4>), gridd, dim3(256), 0, s, b, nx, ny);
      else
#ifdef MPN_BF16_ABLATE
#define MPN_BDIR_ABL(v) if (g_bf16_bdir_abl == v) hipLaunchKernelGGL((conv2d_c8i_bf16_bdir_kernel<3, v>), gridd, dim3(256), 0, s, b, nx, ny); else
      MPN_BDIR_ABL(1) MPN_BDIR_ABL(2) MPN_BDIR_ABL(4) MPN_BDIR_ABL(8) MPN_BDIR_ABL(16) MPN_BDIR_ABL(32) MPN_BDIR_ABL(7) MPN_BDIR_ABL(23) MPN_BDIR_ABL(6) MPN_BDIR_ABL(24) MPN_BDIR_ABL(36) MPN_BDIR_ABL(64) MPN_BDIR_ABL(192) MPN_BDIR_ABL(194) MPN_BDIR_ABL(198) MPN_BDIR_ABL(70) MPN_BDIR_ABL(256)
#undef MPN_BDIR_ABL
#endif
#endif
      hipLaunchKernelGGL((conv2d_c8i_bf16_bdir_kernel<3>), gridd, dim3(256), 0, s, b, nx, ny);
      MPN_CHECK_LAUNCH();
      return MPN_OK;
    }
    // large layers: the LDS-DMA kernel (32-channel stages, 32-bit record offsets); tile shape per layer
    if (g_bf16_dma && b.nch2 % 4 == 0 && b.P >= (g_bf16_dma == 2 ? 1 : 256 * 128) && (size_t)in.B * in.H * in.W * 16 < ((size_t)1 << 32)) {
      // cost of a shape = block-rounds over the 256 CUs x work per block (the two 3-ring shapes run two blocks per CU at half speed
      // each: the same count); 256-cout tiles need whole 256-row weight tiles
      struct Shape { int tm, tn; };
      const Shape shapes[3] = {{256, 128}, {128, 256}, {256, 256}};  // ties go to the earlier entry
#ifdef MPN_DEBUG_HOOKS
      const bool w8 = g_bf16_dma_tn == 2256 && b.CoutP % 256 == 0;  // (hook) the 8-wave 256 x 256 shape: bit-identical, measured no faster than the per-layer pick (ResNet towers 1.95 vs 1.89 ms: 1.5 block rounds on 49 000 pixels), debug flavour only
#else
      constexpr bool w8 = false;
#endif
      int best = -1; long long best_cost = 0;
      for (int i = 0; i < 3; ++i) {
        const Shape &sh = shapes[i];
        if (sh.tm == 256 && b.CoutP % 256 != 0) continue;
        if (w8 && !(sh.tm == 256 && sh.tn == 256)) continue;
        if (!w8 && g_bf16_dma_tn && !(sh.tn == g_bf16_dma_tn % 1000 && sh.tm == (g_bf16_dma_tn >= 1000 ? 128 : 256))) continue;
        const long long nb = (b.P + sh.tn - 1) / sh.tn * (b.CoutP / sh.tm);
        const long long cost = (nb + 255) / 256 * sh.tm * sh.tn;
        if (best < 0 || cost < best_cost) { best = i; best_cost = cost; }
      }
      if (best >= 0) {
        const int tm = shapes[best].tm, tn = shapes[best].tn;
#ifdef MPN_DEBUG_HOOKS
        const bool nch8 = g_bf16_nch == 8 && b.nch2 % 8 == 0;
#else
        constexpr bool nch8 = false;  // 64-channel stages measured 12 % SLOWER (one block per CU: tools/bench_conv_bf16.py, profiles/r04_bf16_conv_ablation.txt)
#endif
        const int ring = nch8 ? ((tm == 256 && tn == 256) ? 2 : 3) : ((tm == 256 && tn == 256) ? 4 : 3);
        const size_t LDS = (size_t)ring * (nch8 ? 8 : 4) * (tm + tn) * 16;  // ring depth x stage bytes (as in the kernel)
        {
          int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<4, 4>), 4 * 32768);
#ifdef MPN_DEBUG_HOOKS
          if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<4, 2, 4, 4>), 4 * 32768);
#endif
          if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<4, 2>), 3 * 24576);
          if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<2, 4>), 3 * 24576);
#ifdef MPN_DEBUG_HOOKS
          if (rc_attr == MPN_OK && nch8) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<4, 4, 8>), 2 * 65536);
          if (rc_attr == MPN_OK && nch8) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<4, 2, 8>), 3 * 49152);
          if (rc_attr == MPN_OK && nch8) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_bf16_dma_kernel<2, 4, 8>), 3 * 49152);
#endif
          if (rc_attr) return rc_attr;
        }
        const int nx = (int)((b.P + tn - 1) / tn), ny = b.CoutP / tm;
        b.trace = (g_bf16_trace && b.KH == g_bf16_trace_kh) ? g_bf16_trace : nullptr;
        b.exp = g_bf16_exp;
        const dim3 gridd((unsigned)(((nx + 7) / 8) * 8 * ny));
#ifdef MPN_DEBUG_HOOKS
        if (nch8 && tm == 256 && tn == 128) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<4, 2, 8>), gridd, dim3(256), LDS, s, b, nx, ny);
        else if (nch8 && tm == 128) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<2, 4, 8>), gridd, dim3(256), LDS, s, b, nx, ny);
        else if (nch8) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<4, 4, 8>), gridd, dim3(256), LDS, s, b, nx, ny);
        else
#endif
#ifdef MPN_DEBUG_HOOKS
        if (w8) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<4, 2, 4, 4>), gridd, dim3(512), LDS, s, b, nx, ny);
        else
#endif
        if (tm == 256 && tn == 128) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<4, 2>), gridd, dim3(256), LDS, s, b, nx, ny);
        else if (tm == 128) hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<2, 4>), gridd, dim3(256), LDS, s, b, nx, ny);
        else hipLaunchKernelGGL((conv2d_c8i_bf16_dma_kernel<4, 4>), gridd, dim3(256), LDS, s, b, nx, ny);
        MPN_CHECK_LAUNCH();
        return MPN_OK;
      }
    }
    dim3 grid((unsigned)(((b.P + 127) / 128 + 7) / 8 * 8 * (b.CoutP / 128)));  // pixel tiles rounded up to the 8 XCDs x cout tiles (see the kernel)
    const long long n_tiles = (b.P + 127) / 128 * (b.CoutP / 128);
    // small layers (layer2 / layer3 of the trunk: a few dozen tiles for 256 CUs): split K across blockIdx.z into fp32 slabs
    {
      const int kp = b.nch2 % 4 == 0 ? 2 : 1;
      const int nstages = b.KH * b.KW * (b.nch2 / (2 * kp));
      const long long nblocks = n_tiles;
      const size_t slab = (size_t)b.CoutP * o->pitch() * sizeof(float);
      int want = (int)std::min<long long>(nstages / 2, (g_bf16_split_target + nblocks - 1) / nblocks);
      if (c.ws && slab) want = (int)std::min<size_t>((size_t)want, c.ws_bytes / slab); else want = 1;
      if (g_bf16_split_target > 0 && !inv && nblocks < g_split_max_tiles && want >= 2) {
        b.stages_per_split = (nstages + want - 1) / want;
        const int splits = (nstages + b.stages_per_split - 1) / b.stages_per_split;
        b.part = c.ws;
        grid.z = (unsigned)splits;
        if (kp == 2) hipLaunchKernelGGL((conv2d_c8i_bf16_kernel<2>), grid, dim3(256), 0, s, b);
        else hipLaunchKernelGGL((conv2d_c8i_bf16_kernel<1>), grid, dim3(256), 0, s, b);
        MPN_CHECK_LAUNCH();
        const size_t total = (size_t)b.Cb_out * (size_t)b.P * 2;
        hipLaunchKernelGGL(conv_splitk_finalize_bf16_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, c.ws, splits, b.CoutP / 8, b.Cb_out, b.pitch_out,
                           b.P, b.bpk, b.res, b.relu, b.out, b.norelu_cb0, b.norelu_cb1);
        MPN_CHECK_LAUNCH();
        return MPN_OK;
      }
    }
    // 32-channel stages (32 KiB of LDS, 4-5 blocks per CU) measured 3 % faster than 64-channel ones on ResNet-50; either way this
    // kernel is bound by its operand loads (64 FLOP per loaded byte at a 128 x 128 tile), not by the bf16 matrix pipe
    if (b.nch2 % 4 == 0) hipLaunchKernelGGL((conv2d_c8i_bf16_kernel<2>), grid, dim3(256), 0, s, b);
    else hipLaunchKernelGGL((conv2d_c8i_bf16_kernel<1>), grid, dim3(256), 0, s, b);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  GConvArgs a{};
  a.in = in.p; a.wpk = c.wpk; a.bpk = c.bpk; a.res = res; a.out = out;
  a.B = in.B; a.Cb_in = in.Cb(); a.H = in.H; a.W = in.W; a.nch = (c.Cin + 7) / 8;
  a.CoutP = round_up(c.Cout, 128); a.Cb_out = (c.Cout + 7) / 8;
  a.KH = c.KH; a.KW = c.KW; a.sh = c.sh; a.sw = c.sw; a.ph = c.ph; a.pw = c.pw;
  a.OH = (in.H + 2 * c.ph - c.KH) / c.sh + 1; a.OW = (in.W + 2 * c.pw - c.KW) / c.sw + 1;
  a.relu = relu; a.norelu_cb0 = c.norelu_c0 / 8; a.norelu_cb1 = c.norelu_c1 / 8;
  MPN_CHECK_ARG(in.C == c.Cin && a.OH > 0 && a.OW > 0);
  a.P = (long long)in.B * a.OH * a.OW;
  *o = ActI{out, in.B, c.Cout, a.OH, a.OW};
  a.pitch_in = in.pitch(); a.pitch_out = o->pitch();
  // Row invariance on the mosaic: the F(2x2,3x3) kernel's rounding depends on a pixel's position inside its 2x2 output tile, and ROI b
  // sits at ((b / mx) * (H + 1) + 1, (b % mx) * (W + 1) + 1) — the same tile phase for every b only when both cell pitches are EVEN
  // (7x7 maps: pitch 8).  Even map sizes (pooled 16 -> 8x8 maps, pitch 9) would make a row depend on its index in the batch: those take
  // the generic kernel whenever rows must not depend on the batch (inv).
  const bool mos_phase_ok = ((in.H + 1) % 2 == 0) && ((in.W + 1) % 2 == 0);
  if (c.wino && c.mos_in && !res && (inv || in.B > 1) && (!inv || mos_phase_ok) && in.H == c.mos_h && in.W == c.mos_w && c.norelu_c1 == c.norelu_c0 && (g_graph_fuse & 128) &&
      ((in.B + c.mos_mx - 1) / c.mos_mx) * (in.H + 1) <= c.mos_rows) {
    // per-ROI 3x3 / stride-1 convolution (layer4's conv2 of blocks 2, 3): the batch as a mosaic image on the Winograd kernel
    const int rows = ((in.B + c.mos_mx - 1) / c.mos_mx) * (in.H + 1);
    const int Hp = act_hp(c.mos_rows), Wp = act_wp(c.mos_cols);
    const size_t recs = (size_t)in.B * in.H * in.W;
    hipLaunchKernelGGL(c8i_to_mosaic_kernel, dim3((unsigned)cdiv_sz(recs * in.Cb() * 2, 256)), dim3(256), 0, s, in.p, in.Cb(), in.B, in.H, in.W, in.pitch(), c.mos_mx, Hp, Wp,
                       c.mos_in);
    MPN_CHECK_LAUNCH();
    const Act ain{c.mos_in, c.Cin, rows, c.mos_cols, Hp, Wp}, aout{c.mos_out, c.Cout, rows, c.mos_cols, Hp, Wp};
    int rcw = conv3x3_c8p(ain, nullptr, c.bpk, c.Cout, relu, aout, Act{nullptr, 0, 0, 0, 0, 0}, s, c.wino, inv);
    if (rcw) return rcw;
    hipLaunchKernelGGL(mosaic_to_c8i_kernel, dim3((unsigned)cdiv_sz(recs * o->Cb() * 2, 256)), dim3(256), 0, s, c.mos_out, o->Cb(), in.B, in.H, in.W, c.mos_mx, Hp, Wp,
                       o->pitch(), out);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  if (allow_gemm && c.col_w && in.planar && in.B == 1 && in.C == 3 && !res && c.norelu_c1 == c.norelu_c0 && (g_graph_fuse & 16)) {
    const int Kc = c.KH * c.KW * 3, nkb = round_up(Kc, 64) / 8;
    const size_t need = (size_t)nkb * o->pitch() * 8 * sizeof(float);
    void *col = nullptr;
    { int rc_ws = scratch_get(SCR_IM2COL, need, s, &col); if (rc_ws) return rc_ws; }
    hipLaunchKernelGGL(im2col3_c8i_kernel, dim3((unsigned)cdiv_sz((size_t)a.P * nkb, 256)), dim3(256), 0, s, in.planar, in.H, in.W, c.KH, c.KW, c.sh, c.sw, c.ph, c.pw,
                       a.OH, a.OW, nkb, o->pitch(), static_cast<float *>(col));
    MPN_CHECK_LAUNCH();
    return linear_c8(static_cast<const float *>(col), (int)a.P, Kc, c.col_w, c.col_b, c.Cout, relu, out, nullptr, s, (int)o->pitch());
  }
  // a pointwise convolution on 1x1 maps is a fully-connected layer (AlexNet's fc7): few row tiles, so the GEMM's split-K form
  // with its row-invariant segments rather than a 128-pixel-tile convolution (the un-split form folds at the same segments)
  if (allow_gemm && c.lin_w && c.norelu_c1 == c.norelu_c0 && !res && in.H == 1 && in.W == 1 && (g_graph_fuse & 8) &&
      (inv || !linear_c8_is_direct((int)in.rows(), c.Cout, (int)in.pitch())))
    return linear_c8(in.p, (int)in.rows(), c.Cin, c.lin_w, c.lin_b, c.Cout, relu, out, nullptr, s, (int)in.pitch(), nullptr, 1);
  if (allow_gemm && c.lin_w && c.norelu_c1 == c.norelu_c0 && (inv || linear_c8_is_direct((int)in.rows(), c.Cout, (int)in.pitch())))  // same rows in and out: the tuned GEMM, residual + ReLU fused
    return linear_c8(in.p, (int)in.rows(), c.Cin, c.lin_w, c.lin_b, c.Cout, relu, out, nullptr, s, (int)in.pitch(), res, inv ? 2 : 0);
  dim3 grid((unsigned)(((a.P + 127) / 128 + 7) / 8 * 8 * (a.CoutP / 128)));  // pixel tiles rounded up to the 8 XCDs x cout tiles (see the kernel)
  // 32-channel stages: the LDS-DMA / hand-pipelined kernel (32-bit gather offsets: the input batch must stay under 4 GiB)
  constexpr size_t PF_LDS = (size_t)2 * 2 * 4 * 128 * 8 * sizeof(float);
  const bool pf_ok = g_fp32_pf && (size_t)in.B * in.H * in.W * 32 < ((size_t)1 << 32);
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(conv2d_c8i_pf_kernel), (int)PF_LDS); if (rc_attr) return rc_attr; }
  {  // small layers: split K across blockIdx.z into fp32 slabs (as the bf16 graph does)
    const int kc = a.nch % 4 == 0 ? 4 : a.nch % 3 == 0 ? 3 : a.nch % 2 == 0 ? 2 : 1;  // 8-channel chunks per stage
    const int nstages = a.KH * a.KW * (a.nch / kc);
    const long long n_tiles = (a.P + 127) / 128 * (a.CoutP / 128);
    const size_t slab = (size_t)a.CoutP * o->pitch() * sizeof(float);
    int want = (int)std::min<long long>(nstages / 2, (g_bf16_split_target + n_tiles - 1) / n_tiles);
    if (c.ws && slab) want = (int)std::min<size_t>((size_t)want, c.ws_bytes / slab); else want = 1;
    if (g_bf16_split_target > 0 && !inv && n_tiles < g_split_max_tiles && want >= 2) {
      a.stages_per_split = (nstages + want - 1) / want;
      const int splits = (nstages + a.stages_per_split - 1) / a.stages_per_split;
      a.part = c.ws;
      grid.z = (unsigned)splits;
      if (kc == 4 && pf_ok) hipLaunchKernelGGL(conv2d_c8i_pf_kernel, grid, dim3(256), PF_LDS, s, a);
      else if (kc == 4) hipLaunchKernelGGL((conv2d_c8i_kernel<4>), grid, dim3(256), 0, s, a);
      else if (kc == 3) hipLaunchKernelGGL((conv2d_c8i_kernel<3>), grid, dim3(256), 0, s, a);
      else if (kc == 2) hipLaunchKernelGGL((conv2d_c8i_kernel<2>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((conv2d_c8i_kernel<1>), grid, dim3(256), 0, s, a);
      MPN_CHECK_LAUNCH();
      const size_t total = (size_t)a.Cb_out * (size_t)a.P * 2;
      hipLaunchKernelGGL(conv_splitk_finalize_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, c.ws, splits, a.CoutP / 8, a.Cb_out, a.pitch_out, a.P,
                         a.bpk, a.res, a.relu, a.out, a.norelu_cb0, a.norelu_cb1);
      MPN_CHECK_LAUNCH();
      return MPN_OK;
    }
  }
  if (a.nch % 4 == 0 && pf_ok) hipLaunchKernelGGL(conv2d_c8i_pf_kernel, grid, dim3(256), PF_LDS, s, a);
  else if (a.nch % 4 == 0) hipLaunchKernelGGL((conv2d_c8i_kernel<4>), grid, dim3(256), 0, s, a);
  else if (a.nch % 3 == 0) hipLaunchKernelGGL((conv2d_c8i_kernel<3>), grid, dim3(256), 0, s, a);
  else if (a.nch % 2 == 0) hipLaunchKernelGGL((conv2d_c8i_kernel<2>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv2d_c8i_kernel<1>), grid, dim3(256), 0, s, a);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// one residual block: y = relu(conv_n(... relu(conv_1(x))) + shortcut(x)); buffers b[0..3] rotate, x may live in any of them
static int rn_block(const RnBlock &blk, ActI x, float *const bufs[4], hipStream_t s, ActI *out, bool per_roi = false) {
  // pick three scratch buffers different from x.p
  float *free_b[3];
  int nf = 0;
  for (int i = 0; i < 4 && nf < 3; ++i)
    if (bufs[i] != x.p) free_b[nf++] = bufs[i];
  MPN_CHECK_ARG(nf == 3);
  ActI sc = x;
  int rc;
  if (blk.has_sc) {
    rc = rn_conv(blk.sc, x, free_b[2], nullptr, 0, s, &sc, true, per_roi);
    if (rc) return rc;
  }
  ActI y = x;
  const int n = (int)blk.convs.size();
  for (int i = 0; i < n; ++i) {
    const bool last = i == n - 1;
    float *dst = free_b[i & 1];
    if (dst == y.p) dst = free_b[(i & 1) ^ 1];
    ActI o;
    rc = rn_conv(blk.convs[i], y, dst, last ? sc.p : nullptr, 1, s, &o, true, per_roi);
    if (rc) return rc;
    if (last) MPN_CHECK_ARG(o.C == sc.C && o.H == sc.H && o.W == sc.W);
    y = o;
  }
  *out = y;
  return MPN_OK;
}

static void rn_shape(const RnConv &c, int &h, int &w) {
  h = (h + 2 * c.ph - c.KH) / c.sh + 1;
  w = (w + 2 * c.pw - c.KW) / c.sw + 1;
}

int resnet_build(const mpn_resnet_weights *rw, int max_h, int max_w, int max_rois, int pooled, ResNetGraph **out) {
  MPN_CHECK_ARG(rw && out && rw->n_convs > 0 && rw->n_blocks > 0 && rw->n_trunk_blocks > 0 && rw->n_trunk_blocks < rw->n_blocks);
  MPN_CHECK_ARG(rw->w && rw->cin && rw->cout && rw->ksize && rw->stride && rw->pad && rw->block_n_convs && rw->block_has_shortcut);
  ResNetGraph *g = new ResNetGraph();
  g->pooled = pooled; g->max_rois = max_rois;
  g->bf16 = rw->bf16 != 0;
  int rc = MPN_OK;
#define RTRY(x) do { rc = (x); if (rc != MPN_OK) { resnet_free(g); return rc; } } while (0)
  const int n_heads = rw->n_heads > 1 ? rw->n_heads : 1;
  const int n_head_blocks = rw->n_blocks - rw->n_trunk_blocks;
  if (n_heads > 8 || n_head_blocks % n_heads != 0) { set_error("mpn_resnet_create: %d head blocks do not split into %d towers", n_head_blocks, n_heads); resnet_free(g); return MPN_EINVAL; }
  const int per_head = n_head_blocks / n_heads;
  g->heads.resize(n_heads);
  int ci = 0;
  bool head_conv = false;  // set while the head blocks' convolutions are taken
  auto take = [&](RnConv &c) -> int {
    if (ci >= rw->n_convs) { set_error("mpn_resnet_create: block table needs more than %d convolutions", rw->n_convs); return MPN_EINVAL; }
    c.Cin = rw->cin[ci]; c.Cout = rw->cout[ci]; c.K = rw->ksize[ci]; c.stride = rw->stride[ci]; c.pad = rw->pad[ci];
    c.KH = c.KW = c.K; c.sh = c.sw = c.stride; c.ph = c.pw = c.pad;
    if (!(c.Cin > 0 && c.Cout > 0 && c.K > 0 && c.stride > 0 && c.pad >= 0 && rw->w[ci])) { set_error("mpn_resnet_create: bad convolution %d", ci); return MPN_EINVAL; }
    int r = rn_pack(g, c, rw->w[ci], rw->b ? rw->b[ci] : nullptr);
    if (r == MPN_OK && head_conv && !g->bf16 && (g_graph_fuse & 128) && c.K == 3 && c.stride == 1 && c.pad == 1 && c.Cin % 8 == 0 && c.Cout % 8 == 0 && c.Cin >= 16) {
      r = rn_alloc(g, &c.wino, conv_wino_elems(c.Cin, c.Cout) * sizeof(float));
      if (r == MPN_OK) r = pack_conv_weights_wino(rw->w[ci], c.Cin, c.Cout, c.wino, nullptr);
    }
    ++ci;
    return r;
  };
  RTRY(take(g->conv1));
  for (int b = 0; b < rw->n_blocks; ++b) {
    RnBlock blk;
    head_conv = b >= rw->n_trunk_blocks;
    for (int k = 0; k < rw->block_n_convs[b]; ++k) { RnConv c; RTRY(take(c)); blk.convs.push_back(c); }
    blk.has_sc = rw->block_has_shortcut[b] != 0;
    if (blk.has_sc) RTRY(take(blk.sc));
    if (b < rw->n_trunk_blocks) g->trunk.push_back(blk);
    else g->heads[(size_t)(b - rw->n_trunk_blocks) / per_head].push_back(blk);
  }
  if (ci != rw->n_convs) { set_error("mpn_resnet_create: %d convolutions given, the block table uses %d", rw->n_convs, ci); resnet_free(g); return MPN_EINVAL; }
  // shapes at the largest image -> buffer sizes
  int h = max_h, w = max_w;
  size_t te = c8i_elems(1, 8, h, w);
  rn_shape(g->conv1, h, w);
  te = std::max(te, c8i_elems(1, g->conv1.Cout, h, w));
  h = (h + 2 - 3) / 2 + 1; w = (w + 2 - 3) / 2 + 1;  // max-pool 3x3/2 pad 1
  int c = g->conv1.Cout;
  for (auto &blk : g->trunk) {
    int bh = h, bw = w;
    for (auto &cv : blk.convs) {
      if (cv.Cin != c && &cv == &blk.convs[0]) { set_error("mpn_resnet_create: channel mismatch in the trunk"); resnet_free(g); return MPN_EINVAL; }
      rn_shape(cv, bh, bw);
      te = std::max(te, c8i_elems(1, cv.Cout, bh, bw));
      c = cv.Cout;
    }
    h = bh; w = bw;
  }
  g->feat_c = c;
  h = w = pooled;
  int mos_h = 0, mos_w = 0, mos_cin = 0, mos_cout = 0;
  size_t he = c8i_elems(max_rois, c, h, w);
  for (auto &hd : g->heads) {
    int hh = h, hw = w, hc = g->feat_c;
    for (auto &blk : hd) {
      int bh = hh, bw = hw;
      for (auto &cv : blk.convs) {
        if (cv.wino) {  // mosaic geometry of this convolution's input maps (one geometry per graph: the first eligible one's)
          if (mos_h == 0) { mos_h = bh; mos_w = bw; }
          if (bh == mos_h && bw == mos_w) { mos_cin = std::max(mos_cin, cv.Cin); mos_cout = std::max(mos_cout, cv.Cout); cv.mos_h = bh; cv.mos_w = bw; }
          else cv.wino = nullptr;
        }
        rn_shape(cv, bh, bw);
        he = std::max(he, c8i_elems(max_rois, cv.Cout, bh, bw));
        hc = cv.Cout;
      }
      hh = bh; hw = bw;
    }
    if (g->out_c && g->out_c != hc) { set_error("mpn_resnet_create: towers end in different channel counts"); resnet_free(g); return MPN_EINVAL; }
    g->out_c = hc;
  }
  g->tb_elems = te; g->hb_elems = he;
  if (mos_h > 0) {  // the two mosaic images, zeroed once (the converters never touch the zero rows / columns between the maps)
    const int mx = std::max(1, 32 / (mos_w + 1)), rows = ((max_rois + mx - 1) / mx) * (mos_h + 1), cols = mx * (mos_w + 1);
    float *mi = nullptr, *mo = nullptr;
    RTRY(rn_alloc(g, &mi, act_bytes(mos_cin, rows, cols)));
    RTRY(rn_alloc(g, &mo, act_bytes(mos_cout, rows, cols)));
    if (hipMemset(mi, 0, act_bytes(mos_cin, rows, cols)) != hipSuccess || hipMemset(mo, 0, act_bytes(mos_cout, rows, cols)) != hipSuccess) { resnet_free(g); return MPN_EHIP; }
    float *mi2 = mi, *mo2 = mo;
#ifdef MPN_DEBUG_HOOKS
    if (g->heads.size() > 1) {  // towers 1, 3, .. run on the second lane (resnet_head_forward): their own mosaic images
      mi2 = mo2 = nullptr;
      RTRY(rn_alloc(g, &mi2, act_bytes(mos_cin, rows, cols)));
      RTRY(rn_alloc(g, &mo2, act_bytes(mos_cout, rows, cols)));
      if (hipMemset(mi2, 0, act_bytes(mos_cin, rows, cols)) != hipSuccess || hipMemset(mo2, 0, act_bytes(mos_cout, rows, cols)) != hipSuccess) { resnet_free(g); return MPN_EHIP; }
    }
#endif
    for (size_t hi = 0; hi < g->heads.size(); ++hi)
      for (auto &blk : g->heads[hi])
        for (auto &cv : blk.convs)
          if (cv.wino) { cv.mos_in = (hi & 1) ? mi2 : mi; cv.mos_out = (hi & 1) ? mo2 : mo; cv.mos_mx = mx; cv.mos_rows = rows; cv.mos_cols = cols; }
  }
  const size_t esz = g->bf16 ? sizeof(bf16_t) : sizeof(float);
  RTRY(rn_alloc(g, &g->img, c8i_elems(1, 16, max_h, max_w) * esz));
  if (g->conv1.col_w) RTRY(rn_alloc(g, &g->img_planar, (size_t)3 * max_h * max_w * sizeof(float)));
  for (int i = 0; i < 4; ++i) { RTRY(rn_alloc(g, &g->tb[i], te * esz)); MPN_CHECK_HIP(hipMemset(g->tb[i], 0, te * esz)); }
  for (int i = 0; i < 4; ++i) { RTRY(rn_alloc(g, &g->hb[i], he * esz)); MPN_CHECK_HIP(hipMemset(g->hb[i], 0, he * esz)); }  // pad planes must hold finite values
#ifdef MPN_DEBUG_HOOKS
  if (g->heads.size() > 1) {  // the second lane's rotating buffers
    for (int i = 0; i < 4; ++i) { RTRY(rn_alloc(g, &g->hb2[i], he * esz)); MPN_CHECK_HIP(hipMemset(g->hb2[i], 0, he * esz)); }
    g->has_lane2 = true;
  }
#endif
#undef RTRY
  MPN_CHECK_HIP(hipDeviceSynchronize());
  *out = g;
  return MPN_OK;
}

void resnet_free(ResNetGraph *g) {
  if (!g) return;
  for (void *q : g->allocs) (void)hipFree(q);
  delete g;
}

void resnet_set_roi_bins(ResNetGraph *g, int bin_rule) { g->roi_bins = bin_rule; }
int resnet_feat_channels(const ResNetGraph *g) { return g->feat_c; }
int resnet_out_channels(const ResNetGraph *g) { return g->out_c; }
int resnet_n_heads(const ResNetGraph *g) { return (int)g->heads.size(); }
bool resnet_has_features(const ResNetGraph *g, int H, int W) { return g->feat && g->last_h == H && g->last_w == W; }

// ---- op-list graphs ----------------------------------------------------------------------------------------------------
static void gop_out_dims(const GOp &op, int h, int w, int &oh, int &ow) {
  if (op.kind == 3) { oh = h; ow = w; return; }  // LRN
  oh = (h + 2 * op.ph - op.kh) / op.sh + 1;
  ow = (w + 2 * op.pw - op.kw) / op.sw + 1;
  if (op.kind == 1 && op.ceil_mode) {  // nn.SpatialMaxPooling:ceil() / Caffe: round up, but the last window must start inside the padded input
    oh = (h + 2 * op.ph - op.kh + op.sh - 1) / op.sh + 1;
    ow = (w + 2 * op.pw - op.kw + op.sw - 1) / op.sw + 1;
    if (op.ph > 0 && (oh - 1) * op.sh >= h + op.ph) --oh;
    if (op.pw > 0 && (ow - 1) * op.sw >= w + op.pw) --ow;
  }
}

// propagate the spatial dims of one op list from tensor 0's; checks that concatenated writers agree
static int graph_dims(const std::vector<GOp> &ops, std::vector<GTensor> &ts, int h0, int w0) {
  for (auto &t : ts) t.H = t.W = 0;
  ts[0].H = h0; ts[0].W = w0;
  for (const GOp &op : ops) {
    const GTensor &src = ts[op.src];
    if (src.H <= 0) { set_error("graph: op reads tensor %d before it is written", op.src); return MPN_EINVAL; }
    int oh, ow;
    gop_out_dims(op, src.H, src.W, oh, ow);
    if (oh <= 0 || ow <= 0) { set_error("graph: op on tensor %d (%dx%d) has an empty output", op.src, src.H, src.W); return MPN_EINVAL; }
    GTensor &dst = ts[op.dst];
    if (dst.H == 0) { dst.H = oh; dst.W = ow; }
    else if (dst.H != oh || dst.W != ow) { set_error("graph: writers of tensor %d disagree on its size", op.dst); return MPN_EINVAL; }
    for (auto &t : ts)
      if (t.alias_of == op.dst) { t.H = oh; t.W = ow; }
  }
  return MPN_OK;
}

static int graph_parse(ResNetGraph *g, int n_ops, const mpn_graph_op *ops_in, int n_t, const int *tc, std::vector<GOp> &out, std::vector<GTensor> &ts,
                       bool head = false) {
  MPN_CHECK_ARG(n_ops > 0 && ops_in && n_t > 1 && tc);
  ts.resize(n_t);
  for (int i = 0; i < n_t; ++i) { ts[i].C = tc[i]; MPN_CHECK_ARG(tc[i] > 0); }
  const int align = g->bf16 ? 16 : 8;
  for (int i = 0; i < n_t; ++i) { ts[i].alias_of = -1; ts[i].alias_c_off = 0; }
  // Two build-time rewrites of the op list (mpn_debug_set_graph_fuse switches them off); `fused` is the rewritten list.
  std::vector<mpn_graph_op> fused(ops_in, ops_in + n_ops);
  std::vector<char> dead((size_t)n_ops, 0);
  std::vector<std::pair<int, int>> norelu_of((size_t)n_ops, std::make_pair(0, 0));  // fused op -> output channels without ReLU
  auto pointwise_private = [&](const mpn_graph_op &o) {
    return o.kind == 0 && o.kh == 1 && o.kw == 1 && o.sh == 1 && o.sw == 1 && o.ph == 0 && o.pw == 0 && o.dst_c_off == 0 && o.src_c_off == 0 && o.dst > 0 &&
           o.dst < (int)ts.size() && o.src >= 0 && o.src < (int)ts.size() && o.cout == ts[o.dst].C && o.cout % align == 0 && o.w;
  };
  // Commute  average-pool(3x3 / 1, pad 1, count_include_pad) -> pointwise convolution  (Inception's pool branches: 1280 / 2048
  // channels pooled, then reduced to 192): both are linear, so pool(conv(x)) == conv(pool(x)) exactly in real arithmetic, and the
  // pool then runs on the convolution's few output channels instead of its many input channels; the convolution's bias and ReLU
  // move behind the pool.  (Intermediate rounding differs: fp32 sums in another order, bf16 rounds conv(x) instead of pool(x).)
  for (int i = 0; (g_graph_fuse & 2) && i < n_ops; ++i) {
    const mpn_graph_op pi = fused[i];
    if (dead[i] || pi.kind != 2 || pi.kh != 3 || pi.kw != 3 || pi.sh != 1 || pi.sw != 1 || pi.ph != 1 || pi.pw != 1 || pi.dst_c_off != 0 || pi.src_c_off != 0) continue;
    if (pi.dst <= 0 || pi.dst >= n_t || pi.src < 0 || pi.src >= n_t || tc[pi.dst] != tc[pi.src] || ts[pi.src].alias_of >= 0) continue;
    int cons = -1, n_cons = 0, n_writers = 0;
    for (int j = 0; j < n_ops; ++j) {
      if (dead[j]) continue;
      if (fused[j].src == pi.dst) { cons = j; ++n_cons; }
      if (fused[j].dst == pi.dst) ++n_writers;
    }
    if (n_cons != 1 || n_writers != 1 || cons <= i) continue;
    const mpn_graph_op cj = fused[cons];
    if (cj.kind != 0 || cj.kh != 1 || cj.kw != 1 || cj.sh != 1 || cj.sw != 1 || cj.ph != 0 || cj.pw != 0 || !cj.w || cj.cout % align != 0 || cj.src_c_off != 0) continue;
    const int tid_new = (int)ts.size();
    GTensor tt; tt.C = cj.cout;
    ts.push_back(tt);
    float *zb = nullptr;
    int rc = rn_alloc(g, &zb, (size_t)cj.cout * sizeof(float));
    if (rc) return rc;
    MPN_CHECK_HIP(hipMemset(zb, 0, (size_t)cj.cout * sizeof(float)));
    mpn_graph_op conv = cj;   // the convolution first: same weights, no bias, no ReLU, on the pool's input
    conv.src = pi.src; conv.dst = tid_new; conv.dst_c_off = 0; conv.relu = 0; conv.b = zb;
    mpn_graph_op pool = pi;   // then the pool, into the convolution's destination, + its bias and ReLU
    pool.src = tid_new; pool.dst = cj.dst; pool.dst_c_off = cj.dst_c_off; pool.cin = cj.cout; pool.relu = cj.relu; pool.b = cj.b ? cj.b : zb;
    fused[i] = conv;
    fused[cons] = pool;
  }
  // Sibling fusion: pointwise (1x1 / stride 1) convolutions that read the SAME tensor and each own a whole private tensor
  // (Inception's branch stems: Mixed_7a's two 768 -> 192, Mixed_7b/7c's 1280 -> 384 and -> 448) become ONE convolution whose
  // output tensor holds their channels side by side; the original tensors become channel-plane views of it (a plane offset in
  // this layout).  The big operand — the per-ROI activation batch, 0.3-0.9 GB — is then read once instead of once per branch,
  // and 192 + 192 couts fill three 128-wide tiles instead of four.  It runs second, so that the commuted pool branches'
  // convolutions (no bias, no ReLU, same input) join their module's group: ReLU-less members go last and the fused convolution
  // skips the ReLU on their channel range (RnConv.norelu_c0 / c1).
  for (int i = 0; (g_graph_fuse & 1) && i < n_ops; ++i) {
    if (dead[i] || !pointwise_private(fused[i])) continue;
    std::vector<int> grp{i};
    for (int j = i + 1; j < n_ops; ++j) {
      if (!dead[j] && fused[j].dst == fused[i].src) break;  // the shared input is rewritten: stop looking
      if (!dead[j] && pointwise_private(fused[j]) && fused[j].src == fused[i].src && fused[j].cin == fused[i].cin && fused[j].dst != fused[i].dst)
        grp.push_back(j);
    }
    if (grp.size() < 2) continue;
    std::stable_sort(grp.begin(), grp.end(), [&](int x, int y) { return fused[x].relu > fused[y].relu; });  // ReLU members first
    int ctot = 0, c_norelu = -1;
    for (int j : grp) {
      if (!fused[j].relu && c_norelu < 0) c_norelu = ctot;
      ctot += fused[j].cout;
    }
    const int cin = fused[i].cin;
    float *wsum = nullptr, *bsum = nullptr;
    int rc = rn_alloc(g, &wsum, (size_t)ctot * cin * sizeof(float));
    if (rc == MPN_OK) rc = rn_alloc(g, &bsum, (size_t)ctot * sizeof(float));
    if (rc) return rc;
    MPN_CHECK_HIP(hipMemset(bsum, 0, (size_t)ctot * sizeof(float)));
    const int fid = (int)ts.size();
    GTensor ft; ft.C = ctot;
    int off = 0;
    for (int j : grp) {
      const mpn_graph_op oj = fused[j];
      MPN_CHECK_HIP(hipMemcpy(wsum + (size_t)off * cin, oj.w, (size_t)oj.cout * cin * sizeof(float), hipMemcpyDeviceToDevice));
      if (oj.b) MPN_CHECK_HIP(hipMemcpy(bsum + off, oj.b, (size_t)oj.cout * sizeof(float), hipMemcpyDeviceToDevice));
      ts[oj.dst].alias_of = fid; ts[oj.dst].alias_c_off = off;
      if (j != i) dead[j] = 1;
      off += oj.cout;
    }
    ts.push_back(ft);
    const int any_relu = fused[grp[0]].relu;
    fused[i].cout = ctot; fused[i].dst = fid; fused[i].w = wsum; fused[i].b = bsum; fused[i].relu = any_relu;
    norelu_of[i] = (any_relu && c_norelu >= 0) ? std::make_pair(c_norelu, ctot) : std::make_pair(0, 0);
  }
  const int n_t_all = (int)ts.size();
  for (int i = 0; i < n_ops; ++i) {
    if (dead[i]) continue;
    const mpn_graph_op &o = fused[i];
    MPN_CHECK_ARG(o.kind >= 0 && o.kind <= 3 && o.src >= 0 && o.src < n_t_all && o.dst > 0 && o.dst < n_t_all && o.src != o.dst);
    MPN_CHECK_ARG(o.kh > 0 && o.kw > 0 && o.sh > 0 && o.sw > 0 && o.ph >= 0 && o.pw >= 0 && o.dst_c_off >= 0 && o.dst_c_off % align == 0);
    MPN_CHECK_ARG(o.src_c_off >= 0 && o.src_c_off % align == 0 && o.cin > 0);
    if (o.kind == 3) {
      if (g->bf16) { set_error("graph: op %d: cross-channel LRN is fp32 only", i); return MPN_EINVAL; }
      MPN_CHECK_ARG(o.kh % 2 == 1 && o.kh <= 17 && o.lrn_beta > 0.0f);
    }
    GOp op;
    op.kind = o.kind; op.src = o.src; op.dst = o.dst; op.dst_c_off = o.dst_c_off;
    op.kh = o.kh; op.kw = o.kw; op.sh = o.sh; op.sw = o.sw; op.ph = o.ph; op.pw = o.pw; op.relu = o.relu;
    op.src_c_off = o.src_c_off; op.cin = o.cin; op.ceil_mode = o.ceil_mode;
    op.lrn_alpha = o.lrn_alpha; op.lrn_beta = o.lrn_beta; op.lrn_k = o.lrn_k;
    const int wc = o.kind == 0 ? o.cout : o.cin;  // channels written
    // a channel range inside a wider tensor must consist of whole channel blocks (its last block may be ragged only at the tensor's end)
    if (o.src_c_off + o.cin > ts[o.src].C || (o.src_c_off + o.cin < ts[o.src].C && o.cin % align != 0) || o.dst_c_off + wc > ts[o.dst].C ||
        (o.dst_c_off + wc < ts[o.dst].C && wc % align != 0)) {
      set_error("graph: op %d: channel mismatch (src %d has %d, writes %d at %d of %d)", i, o.src, ts[o.src].C, wc, o.dst_c_off, ts[o.dst].C);
      return MPN_EINVAL;
    }
    if (o.kind == 2 && o.b) {  // commuted pool: a private copy of the convolution's bias, padded to whole channel blocks
      MPN_CHECK_ARG(o.src_c_off == 0 && o.cin == ts[o.src].C);
      const size_t nb = (size_t)round_up(ts[o.src].C, 8);
      int rc = rn_alloc(g, &op.pool_bias, nb * sizeof(float));
      if (rc) return rc;
      MPN_CHECK_HIP(hipMemset(op.pool_bias, 0, nb * sizeof(float)));
      MPN_CHECK_HIP(hipMemcpy(op.pool_bias, o.b, (size_t)ts[o.src].C * sizeof(float), hipMemcpyDeviceToDevice));
    }
    if (o.kind == 0) {
      MPN_CHECK_ARG(o.w && o.cout > 0);
      RnConv &c = op.conv;
      c.Cin = o.cin; c.Cout = o.cout; c.KH = o.kh; c.KW = o.kw; c.sh = o.sh; c.sw = o.sw; c.ph = o.ph; c.pw = o.pw;
      c.K = o.kh; c.stride = o.sh; c.pad = o.ph;
      c.norelu_c0 = norelu_of[i].first; c.norelu_c1 = norelu_of[i].second;
      int rc = rn_pack(g, c, o.w, o.b);
      if (rc) return rc;
      // trunk 3x3 / stride 1 / pad 1 on whole channel blocks: dense.hip's Winograd kernel (2.25x fewer multiplies; the generic
      // kernel is the fallback when a tensor is a fused view)
      if (!head && !g->bf16 && (g_graph_fuse & 32) && o.kh == 3 && o.kw == 3 && o.sh == 1 && o.sw == 1 && o.ph == 1 && o.pw == 1 && o.cin % 8 == 0 &&
          o.cout % 8 == 0 && o.cin >= 16 && ts[o.src].alias_of < 0 && ts[o.dst].alias_of < 0 && norelu_of[i].first == norelu_of[i].second) {
        bool viewed = false;
        for (const GTensor &t : ts) viewed = viewed || t.alias_of == o.src || t.alias_of == o.dst;
        if (!viewed) {
          rc = rn_alloc(g, &c.wino, conv_wino_elems(o.cin, o.cout) * sizeof(float));
          if (rc == MPN_OK) rc = pack_conv_weights_wino(o.w, o.cin, o.cout, c.wino, nullptr);
          if (rc) return rc;
        }
      }
      // A convolution whose window is the whole ROI-pooled map (AlexNet's fc6: alexnet.lua's View(-1) + Linear(9216, 4096)) is a
      // fully-connected layer: with the pooled bins written as (bin, roi) rows it is ONE GEMM over K = Cin * bins on the tuned
      // kernel (the route the VGG pipeline's fc6 takes) instead of a 36-tap convolution on three 128-ROI pixel tiles.  Needs the
      // pooled tensor all to itself (it is then never materialised as maps).
      if (head && !g->bf16 && (g_graph_fuse & 8) && o.src == 0 && o.src_c_off == 0 && o.cin == ts[0].C && o.kh == g->pooled && o.kw == g->pooled && o.ph == 0 &&
          o.pw == 0 && o.dst_c_off == 0 && o.cout == ts[o.dst].C && o.cin % 32 == 0 && (o.cin * o.kh * o.kw) % 64 == 0) {
        bool sole = true;
        for (int j = 0; j < n_ops; ++j)
          if (j != i && !dead[j] && fused[j].src == 0) sole = false;
        if (sole) {
          const int Kfc = o.cin * o.kh * o.kw;
          rc = rn_alloc(g, &op.fc_w, lin_wpk_elems(Kfc, o.cout) * sizeof(float));
          if (rc == MPN_OK) rc = rn_alloc(g, &op.fc_b, (size_t)lin_np(o.cout) * sizeof(float));
          if (rc == MPN_OK) rc = pack_linear_weights(o.w, o.b, Kfc, o.cout, o.kh * o.kw, op.fc_w, op.fc_b, nullptr);
          if (rc) return rc;
        }
      }
    }
    out.push_back(op);
  }
  return MPN_OK;
}

int graph_build(const mpn_graph_weights *gw, int max_h, int max_w, int max_rois, int pooled, ResNetGraph **out) {
  MPN_CHECK_ARG(gw && out && gw->feat_tensor > 0 && gw->feat_tensor < gw->n_trunk_tensors && gw->out_tensor > 0 && gw->out_tensor < gw->n_head_tensors);
  ResNetGraph *g = new ResNetGraph();
  g->is_graph = true; g->pooled = pooled; g->max_rois = max_rois; g->bf16 = gw->bf16 != 0;
  g->feat_tensor = gw->feat_tensor; g->out_tensor = gw->out_tensor;
  int rc = graph_parse(g, gw->n_trunk_ops, gw->trunk_ops, gw->n_trunk_tensors, gw->trunk_tensor_c, g->g_trunk, g->t_trunk);
  const int n_heads = gw->n_heads > 1 ? gw->n_heads : 1;
  MPN_CHECK_ARG(n_heads <= 8);
  g->g_heads.resize(n_heads);
  for (int t = 0; rc == MPN_OK && t < n_heads; ++t) {
    rc = graph_parse(g, gw->n_head_ops, gw->head_ops + (size_t)t * gw->n_head_ops, gw->n_head_tensors, gw->head_tensor_c, g->g_heads[t], g->t_head, true);
    if (rc == MPN_OK)
      for (GOp &op : g->g_heads[t])
        if (op.fc_w && !g->fc_x) {
          const size_t bytes = (size_t)(op.cin / 8) * pooled * pooled * round_up(max_rois, 128) * 8 * sizeof(float);
          rc = rn_alloc(g, &g->fc_x, bytes);
          if (rc == MPN_OK && hipMemset(g->fc_x, 0, bytes) != hipSuccess) rc = MPN_EHIP;
        }
    if (rc == MPN_OK)
      for (GOp &op : g->g_heads[t])
        if (op.kind == 1 && op.src == 0 && op.src_c_off == 0 && op.cin == g->t_head[0].C && op.kh == op.kw && op.sh == op.sw && op.ph == op.pw && !op.ceil_mode)
          op.from_rois = true;
  }
  if (rc == MPN_OK && (g->t_trunk[0].C != 3 || g->t_head[0].C != g->t_trunk[g->feat_tensor].C)) {
    set_error("graph: tensor 0 must be the 3-channel image (trunk) / carry the feature tensor's channels (head)");
    rc = MPN_EINVAL;
  }
  if (rc == MPN_OK) rc = graph_dims(g->g_trunk, g->t_trunk, max_h, max_w);
  for (int t = 0; rc == MPN_OK && t < n_heads; ++t) rc = graph_dims(g->g_heads[t], g->t_head, pooled, pooled);
  const size_t esz = g->bf16 ? sizeof(bf16_t) : sizeof(float);
  for (size_t i = 0; rc == MPN_OK && i < g->t_trunk.size(); ++i) {
    GTensor &t = g->t_trunk[i];
    if (t.H == 0 || t.alias_of >= 0) continue;  // never written / a view of a fused tensor
    const size_t bytes = c8i_elems(1, i == 0 ? 16 : t.C, t.H, t.W) * esz;
    rc = rn_alloc(g, &t.buf, bytes);
    if (rc == MPN_OK && hipMemset(t.buf, 0, bytes) != hipSuccess) rc = MPN_EHIP;
  }
  // a tensor lives in ONE layout at a time: the Winograd route only where every writer of the destination takes it (a DepthConcat
  // of a 3x3 branch with pointwise / pooled branches stays on the generic kernels)
  for (GOp &op : g->g_trunk)
    if (op.conv.wino)
      for (const GOp &other : g->g_trunk)
        if (other.dst == op.dst && !(other.kind == 0 && other.conv.wino)) { op.conv.wino = nullptr; break; }
  for (bool again = true; again;) {  // (clearing one writer may orphan another of the same tensor)
    again = false;
    for (GOp &op : g->g_trunk)
      if (op.conv.wino)
        for (const GOp &other : g->g_trunk)
          if (other.dst == op.dst && !(other.kind == 0 && other.conv.wino)) { op.conv.wino = nullptr; again = true; break; }
  }
  for (const GOp &op : g->g_trunk) {
    if (rc != MPN_OK || !op.conv.wino) continue;
    for (int id : {op.src, op.dst}) {
      GTensor &t = g->t_trunk[id];
      if (t.c8p || rc != MPN_OK) continue;
      const size_t bytes = act_bytes(t.C, t.H, t.W);
      rc = rn_alloc(g, &t.c8p, bytes);
      if (rc == MPN_OK && hipMemset(t.c8p, 0, bytes) != hipSuccess) rc = MPN_EHIP;
      t.p_h = t.H; t.p_w = t.W;
    }
  }
  for (size_t i = 0; rc == MPN_OK && i < g->t_head.size(); ++i) {
    GTensor &t = g->t_head[i];
    if (t.H == 0 || t.alias_of >= 0) continue;
    const size_t bytes = c8i_elems(max_rois, t.C, t.H, t.W) * esz;
    rc = rn_alloc(g, &t.buf, bytes);
    if (rc == MPN_OK && hipMemset(t.buf, 0, bytes) != hipSuccess) rc = MPN_EHIP;
  }
#ifdef MPN_DEBUG_HOOKS  // measured without a gain on the graph towers (pipeline.hip run_detect): the second lane is not built into the product library
  if (rc == MPN_OK && n_heads > 1) {  // the second lane's activations
    g->t_head2 = g->t_head;
    for (size_t i = 0; rc == MPN_OK && i < g->t_head2.size(); ++i) {
      GTensor &t = g->t_head2[i];
      if (t.H == 0 || t.alias_of >= 0) continue;
      const size_t bytes = c8i_elems(max_rois, t.C, t.H, t.W) * esz;
      t.buf = nullptr;
      rc = rn_alloc(g, &t.buf, bytes);
      if (rc == MPN_OK && hipMemset(t.buf, 0, bytes) != hipSuccess) rc = MPN_EHIP;
    }
    g->has_lane2 = rc == MPN_OK;
  }
#endif
  if (rc != MPN_OK) { resnet_free(g); return rc; }
  g->feat_c = g->t_trunk[g->feat_tensor].C;
  g->out_c = g->t_head[g->out_tensor].C;
  g->img = g->t_trunk[0].buf;
  for (const GOp &op : g->g_trunk)
    if (op.kind == 0 && op.src == 0 && op.conv.col_w && !g->img_planar) {
      rc = rn_alloc(g, &g->img_planar, (size_t)3 * max_h * max_w * sizeof(float));
      if (rc != MPN_OK) { resnet_free(g); return rc; }
    }
  g->heads.resize(n_heads);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  *out = g;
  return MPN_OK;
}

// run one op list on a batch of B maps; dims must have been propagated (graph_dims)
static int graph_run(ResNetGraph *g, const std::vector<GOp> &ops, std::vector<GTensor> &ts, int B, hipStream_t s, bool skip_from_rois = false,
                     bool fc_gemm = false) {
  const size_t esz = g->bf16 ? sizeof(bf16_t) : sizeof(float);
  // layout bookkeeping for the Winograd convolutions (trunk only): which form of each tensor holds this run's values
  const bool trunk = &ts == &g->t_trunk;
  std::vector<char> in_i(ts.size(), 1), in_p(ts.size(), 0);
  auto fit_halo = [&](GTensor &t) -> int {  // the padded buffer's zero halo sits where (t.H, t.W) needs it
    if (t.p_h == t.H && t.p_w == t.W) return MPN_OK;
    MPN_CHECK_HIP(hipMemsetAsync(t.c8p, 0, act_bytes(t.C, t.H, t.W), s));
    t.p_h = t.H; t.p_w = t.W;
    return MPN_OK;
  };
  auto need_c8i = [&](int id) -> int {
    if (in_i[id]) return MPN_OK;
    GTensor &t = ts[id];
    const ActI ti{t.buf, 1, t.C, t.H, t.W};
    const Act tp = make_act(t.c8p, t.C, t.H, t.W);
    hipLaunchKernelGGL(c8p_to_c8i_kernel, dim3((unsigned)cdiv_sz((size_t)t.H * t.W * ti.Cb() * 2, 256)), dim3(256), 0, s, t.c8p, ti.Cb(), t.H, t.W, tp.Hp, tp.Wp,
                       ti.pitch(), t.buf);
    MPN_CHECK_LAUNCH();
    in_i[id] = 1;
    return MPN_OK;
  };
  for (const GOp &op : ops) {
    if (skip_from_rois && op.from_rois) continue;  // already produced from the feature map (resnet_head_forward)
    if (trunk && B == 1 && op.kind == 0 && op.conv.wino && (g_graph_fuse & 32) && ts[op.src].c8p && ts[op.dst].c8p) {
      GTensor &ps = ts[op.src], &pd = ts[op.dst];
      int rc = MPN_OK;
      if (!in_p[op.src]) {
        rc = fit_halo(ps);
        if (rc) return rc;
        const ActI ti{ps.buf, 1, ps.C, ps.H, ps.W};
        const Act tp = make_act(ps.c8p, ps.C, ps.H, ps.W);
        hipLaunchKernelGGL(c8i_to_c8p_kernel, dim3((unsigned)cdiv_sz((size_t)ps.H * ps.W * ti.Cb() * 2, 256)), dim3(256), 0, s, ps.buf, ti.Cb(), ps.H, ps.W, ti.pitch(),
                           tp.Hp, tp.Wp, ps.c8p);
        MPN_CHECK_LAUNCH();
        in_p[op.src] = 1;
      }
      if (!in_p[op.dst]) { rc = fit_halo(pd); if (rc) return rc; }
      const Act fs = make_act(ps.c8p, ps.C, ps.H, ps.W), fd = make_act(pd.c8p, pd.C, pd.H, pd.W);
      const Act ain = Act{ps.c8p + (size_t)(op.src_c_off / 8) * fs.plane(), op.cin, ps.H, ps.W, fs.Hp, fs.Wp};
      const Act aout = Act{pd.c8p + (size_t)(op.dst_c_off / 8) * fd.plane(), op.conv.Cout, pd.H, pd.W, fd.Hp, fd.Wp};
      rc = conv3x3_c8p(ain, nullptr, op.conv.bpk, op.conv.Cout, op.relu, aout, Act{nullptr, 0, 0, 0, 0, 0}, s, op.conv.wino);
      if (rc) return rc;
      in_p[op.dst] = 1; in_i[op.dst] = 0;
      continue;
    }
    if (trunk) {
      int rc = need_c8i(op.src);
      if (rc) return rc;
      // a plain op that writes part of a tensor whose other channels so far exist only in the padded form: bring those over first
      if (!in_i[op.dst] && in_p[op.dst]) { rc = need_c8i(op.dst); if (rc) return rc; }
      in_i[op.dst] = 1; in_p[op.dst] = 0;
    }
    GTensor src = ts[op.src];
    if (src.alias_of >= 0) {  // channel-plane view of a fused tensor (same rows, so the same pitch)
      const GTensor &par = ts[src.alias_of];
      const ActI pa{par.buf, B, par.C, par.H, par.W};
      src.buf = reinterpret_cast<float *>(reinterpret_cast<char *>(par.buf) + (size_t)(src.alias_c_off / 8) * pa.pitch() * 8 * esz);
    }
    if (op.src_c_off > 0 || op.cin != src.C) {  // a channel range of the source (grouped convolutions): plane offset, narrower tensor
      const ActI full{src.buf, B, src.C, src.H, src.W};
      src.buf = reinterpret_cast<float *>(reinterpret_cast<char *>(src.buf) + (size_t)(op.src_c_off / 8) * full.pitch() * 8 * esz);
      src.C = op.cin;
    }
    GTensor &dst = ts[op.dst];
    ActI in{src.buf, B, src.C, src.H, src.W};
    if (&ts == &g->t_trunk && op.src == 0 && op.src_c_off == 0) in.planar = g->img_planar;
    const ActI od{dst.buf, B, dst.C, dst.H, dst.W};
    char *outp = reinterpret_cast<char *>(dst.buf) + (size_t)(op.dst_c_off / 8) * od.pitch() * 8 * esz;  // plane offset = the concat
    if (op.kind == 0 && fc_gemm && op.fc_w) {  // fully connected over the (bin, roi)-row pooled matrix (resnet_head_forward wrote it): row-invariant K segments
      int rc = linear_c8(g->fc_x, B, op.cin * op.kh * op.kw, op.fc_w, op.fc_b, op.conv.Cout, op.relu, reinterpret_cast<float *>(outp), nullptr, s, (int)od.pitch(),
                         nullptr, 1);
      if (rc) return rc;
    } else if (op.kind == 0) {
      ActI o;
      // the GEMM writes whole 128-channel panels: only when this op owns them (no neighbouring branch inside the panel)
      const bool own = op.conv.Cout % 128 == 0 || (op.dst_c_off == 0 && op.conv.Cout == dst.C);
#ifdef MPN_DEBUG_HOOKS
      if (!trunk && op.src == 0 && g_tower_knock) g_knock_arm = 1;  // reads the pooled tensor
#endif
      int rc = rn_conv(op.conv, in, reinterpret_cast<float *>(outp), nullptr, op.relu, s, &o, own, !trunk);
#ifdef MPN_DEBUG_HOOKS
      g_knock_arm = 0;
#endif
      if (rc) return rc;
    } else {
      const size_t total = (size_t)in.Cb() * B * dst.H * dst.W * 2;
      const dim3 grid((unsigned)cdiv_sz(total, 256)), grid16((unsigned)cdiv_sz(total / 2, 256));  // fp32: half records, bf16: whole records per thread
      if (op.kind == 3) {
        const size_t rows = (size_t)B * src.H * src.W;
        hipLaunchKernelGGL(lrn_c8i_kernel, dim3((unsigned)cdiv_sz(rows * in.Cb(), 256)), dim3(256), 0, s, src.buf, in.Cb(), src.C, rows, in.pitch(), op.kh,
                           op.lrn_alpha, op.lrn_beta, op.lrn_k, od.pitch(), reinterpret_cast<float *>(outp));
      } else if (op.kind == 1) {
        MPN_CHECK_ARG(op.kh == op.kw && op.sh == op.sw && op.ph == op.pw);
        if (g->bf16)
          hipLaunchKernelGGL(maxpool2d_c8i_bf16_kernel, grid16, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(src.buf), in.Cb(), B, src.H, src.W, in.pitch(),
                             op.kh, op.sh, op.ph, dst.H, dst.W, od.pitch(), reinterpret_cast<bf16_t *>(outp));
        else
          hipLaunchKernelGGL(maxpool2d_c8i_kernel, grid, dim3(256), 0, s, src.buf, in.Cb(), B, src.H, src.W, in.pitch(), op.kh, op.sh, op.ph, dst.H, dst.W,
                             od.pitch(), reinterpret_cast<float *>(outp));
      } else {
        if (g->bf16 && op.sh == 1 && op.sw == 1 && dst.H == src.H && dst.W == src.W && src.H * src.W <= 256) {
          const int mpb = 256 / (src.H * src.W);
          hipLaunchKernelGGL(avgpool2d_c8i_bf16_small_kernel, dim3((unsigned)((B + mpb - 1) / mpb), (unsigned)in.Cb()), dim3(256), 0, s,
                             reinterpret_cast<const bf16_t *>(src.buf), B, src.H, src.W, in.pitch(), op.kh, op.kw, op.ph, op.pw, od.pitch(),
                             reinterpret_cast<bf16_t *>(outp), op.pool_bias, op.relu);
        } else if (g->bf16)
          hipLaunchKernelGGL(avgpool2d_c8i_bf16_kernel, grid16, dim3(256), 0, s, reinterpret_cast<const bf16_t *>(src.buf), in.Cb(), B, src.H, src.W,
                             in.pitch(), op.kh, op.kw, op.sh, op.sw, op.ph, op.pw, dst.H, dst.W, od.pitch(), reinterpret_cast<bf16_t *>(outp), op.pool_bias, op.relu);
        else
          hipLaunchKernelGGL((avgpool2d_c8i_kernel<float>), grid, dim3(256), 0, s, src.buf, in.Cb(), B, src.H, src.W, in.pitch(), op.kh, op.kw, op.sh, op.sw,
                             op.ph, op.pw, dst.H, dst.W, od.pitch(), reinterpret_cast<float *>(outp), op.pool_bias, op.relu);
      }
      MPN_CHECK_LAUNCH();
    }
  }
  if (trunk) { int rc = need_c8i(g->feat_tensor); if (rc) return rc; }  // the ROI pooling reads the C8I map
  return MPN_OK;
}

int resnet_trunk_forward(ResNetGraph *g, const float *d_image, int H, int W, const int *swap, double scale, const double *mean,
                         const double *std, int has_std, hipStream_t s) {
  MPN_CHECK_ARG(g && d_image && H > 0 && W > 0);
  const size_t plane = (size_t)H * W;
  if (g->bf16)
    hipLaunchKernelGGL(image_transform_c8i_bf16_kernel, dim3((unsigned)cdiv_sz(plane, 256)), dim3(256), 0, s, d_image, H, W, swap[0], swap[1], swap[2],
                       scale, mean[0], mean[1], mean[2], has_std ? std[0] : 1.0, has_std ? std[1] : 1.0, has_std ? std[2] : 1.0, has_std,
                       reinterpret_cast<bf16_t *>(g->img), (ActI{g->img, 1, 3, H, W}).pitch());
  else
    hipLaunchKernelGGL(image_transform_c8i_kernel, dim3((unsigned)cdiv_sz(plane, 256)), dim3(256), 0, s, d_image, H, W, swap[0], swap[1], swap[2], scale,
                       mean[0], mean[1], mean[2], has_std ? std[0] : 1.0, has_std ? std[1] : 1.0, has_std ? std[2] : 1.0, has_std, g->img, g->img_planar);
  MPN_CHECK_LAUNCH();
  if (g->is_graph) {
    int rc = graph_dims(g->g_trunk, g->t_trunk, H, W);
    if (rc == MPN_OK) rc = graph_run(g, g->g_trunk, g->t_trunk, 1, s);
    if (rc) return rc;
    const GTensor &f = g->t_trunk[g->feat_tensor];
    g->feat = f.buf; g->feat_h = f.H; g->feat_w = f.W; g->last_h = H; g->last_w = W; g->feat_sorted_valid = false; g->feat_vmax_valid = false;
    return MPN_OK;
  }
  ActI x{g->img, 1, 3, H, W}, y;
  x.planar = g->img_planar;
  int rc = rn_conv(g->conv1, x, g->tb[0], nullptr, 1, s, &y);
  if (rc) return rc;
  const int OH = (y.H + 2 - 3) / 2 + 1, OW = (y.W + 2 - 3) / 2 + 1;
  {
    const size_t total = (size_t)y.Cb() * OH * OW * 2;
    const ActI po{g->tb[1], 1, y.C, OH, OW};
    if (g->bf16)
      hipLaunchKernelGGL(maxpool2d_c8i_bf16_kernel, dim3((unsigned)cdiv_sz(total / 2, 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(y.p), y.Cb(), 1,
                         y.H, y.W, y.pitch(), 3, 2, 1, OH, OW, po.pitch(), reinterpret_cast<bf16_t *>(g->tb[1]));
    else
      hipLaunchKernelGGL(maxpool2d_c8i_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, y.p, y.Cb(), 1, y.H, y.W, y.pitch(), 3, 2, 1, OH, OW,
                         po.pitch(), g->tb[1]);
    MPN_CHECK_LAUNCH();
  }
  ActI cur{g->tb[1], 1, y.C, OH, OW};
  for (auto &blk : g->trunk) {
    rc = rn_block(blk, cur, g->tb, s, &y);
    if (rc) return rc;
    cur = y;
  }
  g->feat = cur.p; g->feat_h = cur.H; g->feat_w = cur.W; g->last_h = H; g->last_w = W; g->feat_sorted_valid = false; g->feat_vmax_valid = false;
  return MPN_OK;
}

// The per-image preparation of the bf16 ROI pooling — the order-preserving int16 image of the feature map and, where a head fuses the
// max-pool of its pooled input, its vertical range-max levels — for head `head` (-1: whatever ANY head needs).  Built once per trunk run
// (the valid flags), on `s`.
static int heads_prepare_impl(ResNetGraph *g, int head, hipStream_t s) {
  const int Cb = (g->feat_c + 7) / 8;
  if (!(g->bf16 && (g_bf16_fast_pool & 1) && Cb % 4 == 0)) return MPN_OK;
  const bool fuse_mp = g->is_graph && (g_graph_fuse & 4) && g->roi_bins == 0;
  const ActI fa{g->feat, 1, g->feat_c, g->feat_h, g->feat_w};
  const size_t need = (size_t)Cb * fa.pitch() * 8;
  if (g->feat_sorted_elems < need) {  // first use (or a larger image than any before): outside the steady state
    float *q = reinterpret_cast<float *>(g->feat_sorted);
    g->feat_sorted = nullptr; g->feat_sorted_elems = 0;
    int rc = rn_regrow(g, &q, need * sizeof(bf16_t));
    if (rc) return rc;
    g->feat_sorted = reinterpret_cast<bf16_t *>(q); g->feat_sorted_elems = need; g->feat_sorted_valid = false;
  }
  bool any_mp = false;
  if (fuse_mp)
    for (int h = 0; h < (int)g->g_heads.size(); ++h)
      if (head < 0 || h == head)
        for (const GOp &op : g->g_heads[h]) any_mp = any_mp || op.from_rois;
  const int want_levels = (any_mp && (g_graph_fuse & 64) && g->feat_h > 1) ? 31 - __builtin_clz((unsigned)g->feat_h) : 0;
  if (want_levels > 0 && (g->feat_vmax_elems < need || g->feat_vmax_levels < want_levels)) {  // outside the steady state, as above
    float *q = reinterpret_cast<float *>(g->feat_vmax);
    g->feat_vmax = nullptr; g->feat_vmax_elems = 0; g->feat_vmax_levels = 0;
    int rc = rn_regrow(g, &q, need * sizeof(bf16_t) * want_levels);
    if (rc) return rc;
    g->feat_vmax = reinterpret_cast<bf16_t *>(q); g->feat_vmax_elems = need; g->feat_vmax_levels = want_levels; g->feat_vmax_valid = false;
  }
  if (!g->feat_sorted_valid) {
    hipLaunchKernelGGL(bf16_sortable_kernel, dim3((unsigned)cdiv_sz(need / 8, 256)), dim3(256), 0, s, reinterpret_cast<const u32x4 *>(g->feat), need / 8,
                       reinterpret_cast<u32x4 *>(g->feat_sorted));
    MPN_CHECK_LAUNCH();
    g->feat_sorted_valid = true;
    g->feat_vmax_valid = false;
  }
  if (want_levels > 0 && !g->feat_vmax_valid) {  // once per image: level l from level l - 1 (level 0 = the sortable map)
    for (int l = 1; l <= want_levels; ++l) {
      const u32x4 *prev = l == 1 ? reinterpret_cast<const u32x4 *>(g->feat_sorted)
                                 : reinterpret_cast<const u32x4 *>(g->feat_vmax) + (size_t)(l - 2) * (g->feat_vmax_elems / 8);
      hipLaunchKernelGGL(vmax_level_sorted_kernel, dim3((unsigned)cdiv_sz((size_t)g->feat_h * g->feat_w * Cb, 256)), dim3(256), 0, s, prev,
                         reinterpret_cast<u32x4 *>(g->feat_vmax) + (size_t)(l - 1) * (g->feat_vmax_elems / 8), g->feat_h, g->feat_w, fa.pitch(), Cb, 1 << (l - 1));
      MPN_CHECK_LAUNCH();
    }
    g->feat_vmax_valid = true;
  }
  return MPN_OK;
}
int resnet_heads_prepare(ResNetGraph *g, hipStream_t s) {
  MPN_CHECK_ARG(g && g->feat);
  return heads_prepare_impl(g, -1, s);
}
bool resnet_has_second_lane(const ResNetGraph *g) { return g->has_lane2 && g_roi_invariant != 0; }  // (per-ROI layers that may split K share one slab workspace: one lane)

int resnet_head_forward(ResNetGraph *g, int head, const float *d_rois, int roi_stride, int N, float spatial_scale, float *d_feat_c8, int Mp,
                        hipStream_t s, int lane) {
  MPN_CHECK_ARG(g && g->feat && d_rois && d_feat_c8 && N > 0 && N <= g->max_rois && head >= 0 && head < (int)g->heads.size());
  MPN_CHECK_ARG(lane == 0 || (lane == 1 && g->has_lane2));
  const int Cb = (g->feat_c + 7) / 8, PH = g->pooled;
  std::vector<GTensor> &t_head = lane ? g->t_head2 : g->t_head;
  float *const *const hb = lane ? g->hb2 : g->hb;
  float *const pool_dst = g->is_graph ? t_head[0].buf : hb[0];
  const bool fuse_mp = g->is_graph && g->bf16 && (g_bf16_fast_pool & 1) && Cb % 4 == 0 && (g_graph_fuse & 4) && g->roi_bins == 0;  // max-pools of the pooled input: from the map (its kernel unions the CUDA branch's bins: the adaptive rule runs the max-pool as an ordinary op on the pooled batch)
  bool fc_gemm = false;  // the head starts with a fully-connected layer: pool straight into its GEMM operand
  if (g->is_graph && !g->bf16 && g->fc_x && (g_graph_fuse & 8) && (g_bf16_fast_pool & 1) && Cb % 4 == 0)
    for (const GOp &op : g->g_heads[head]) fc_gemm = fc_gemm || op.fc_w != nullptr;
  if (g->is_graph) { int rc = graph_dims(g->g_heads[head], t_head, PH, PH); if (rc) return rc; }
  {
    const size_t total = (size_t)N * Cb * PH * PH * 2;
    const ActI fa{g->feat, 1, g->feat_c, g->feat_h, g->feat_w}, pa{pool_dst, N, g->feat_c, PH, PH};
    if (g->bf16 && (g_bf16_fast_pool & 1) && Cb % 4 == 0) {
      { int rc_prep = heads_prepare_impl(g, head, s); if (rc_prep) return rc_prep; }  // (a no-op when resnet_heads_prepare already ran for this image)
      const int want_levels = g->feat_vmax_valid ? g->feat_vmax_levels : 0;
#ifdef MPN_DEBUG_HOOKS
      if (g_tower_knock & 1) {}
      else if ((g_pool_exp & 3) && Cb % 8 == 0) {  // mpn_debug_set_pool_exp (timing experiments): 1 = 8 channel blocks per thread / ordinary stores, 2 = 4 blocks / ordinary stores (rounds 3-5), 3 = 8 blocks / non-temporal
        const dim3 g8((unsigned)cdiv_sz((size_t)N * PH * PH, 256), (unsigned)(Cb / 8)), g4((unsigned)cdiv_sz((size_t)N * PH * PH, 256), (unsigned)(Cb / 4));
#define MPN_POOL_ARGS reinterpret_cast<const u32x4 *>(g->feat_sorted), g->feat_h, g->feat_w, fa.pitch(), d_rois, roi_stride, N, PH, PH, spatial_scale, reinterpret_cast<u32x4 *>(pool_dst), pa.pitch(), g->roi_bins
        if ((g_pool_exp & 3) == 1) hipLaunchKernelGGL((roi_pool_c8i_bf16_sorted_kernel<8, false>), g8, dim3(256), 0, s, MPN_POOL_ARGS);
        else if ((g_pool_exp & 3) == 2) hipLaunchKernelGGL((roi_pool_c8i_bf16_sorted_kernel<4, false>), g4, dim3(256), 0, s, MPN_POOL_ARGS);   // rounds 3-5: ordinary stores
        else hipLaunchKernelGGL((roi_pool_c8i_bf16_sorted_kernel<8, true>), g8, dim3(256), 0, s, MPN_POOL_ARGS);
#undef MPN_POOL_ARGS
      } else
#endif
      // non-temporal stores: the 0.4-0.9 GB pooled tensor streams past the L2 instead of evicting the sorted map the launch gathers from
      // (measured, profiles/r06_pool_exp.txt: configs[4] 24.28 -> 24.10 ms, configs[3] bf16 11.06 -> 10.99; 8 channel blocks per thread: slower)
      hipLaunchKernelGGL((roi_pool_c8i_bf16_sorted_kernel<4, true>), dim3((unsigned)cdiv_sz((size_t)N * PH * PH, 256), (unsigned)(Cb / 4)), dim3(256), 0, s,
                         reinterpret_cast<const u32x4 *>(g->feat_sorted), g->feat_h, g->feat_w, fa.pitch(), d_rois, roi_stride, N, PH, PH, spatial_scale,
                         reinterpret_cast<u32x4 *>(pool_dst), pa.pitch(), g->roi_bins);
      if (fuse_mp)
        for (const GOp &op : g->g_heads[head]) {
          if (!op.from_rois) continue;
          MPN_CHECK_LAUNCH();
          const GTensor &dst = t_head[op.dst];
          const ActI od{dst.buf, N, dst.C, dst.H, dst.W};
          char *outp = reinterpret_cast<char *>(dst.buf) + (size_t)(op.dst_c_off / 8) * od.pitch() * 8 * sizeof(bf16_t);  // plane offset = the concat
          hipLaunchKernelGGL(roi_maxpool_c8i_bf16_sorted_kernel<4>, dim3((unsigned)cdiv_sz((size_t)N * dst.H * dst.W, 256), (unsigned)(Cb / 4)), dim3(256), 0, s,
                             reinterpret_cast<const u32x4 *>(g->feat_sorted), g->feat_h, g->feat_w, fa.pitch(), d_rois, roi_stride, N, PH, PH, spatial_scale,
                             op.kh, op.sh, op.ph, dst.H, dst.W, reinterpret_cast<u32x4 *>(outp), od.pitch(),
                             want_levels > 0 ? reinterpret_cast<const u32x4 *>(g->feat_vmax) : nullptr, g->feat_vmax_elems / 8, want_levels);
        }
    } else if (g->bf16)
      hipLaunchKernelGGL(roi_pool_c8i_bf16_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(g->feat), Cb, g->feat_h,
                         g->feat_w, fa.pitch(), d_rois, roi_stride, N, PH, PH, spatial_scale, reinterpret_cast<bf16_t *>(pool_dst), pa.pitch(), g->roi_bins);
    else if (fc_gemm && (g_graph_fuse & 256)) {
      // the fully-connected operand is the VGG pipeline's (bin, roi)-row matrix: its pooling kernel too (a wave = one (roi, bin) over 256
      // channels of a pixel-major copy of the map, four ROIs per block leaving as whole 128-byte lines) — 37 -> 12 us on AlexNet
      void *pm = nullptr;
      { int rc_ws = scratch_get(SCR_MISC, (size_t)g->feat_h * g->feat_w * Cb * 8 * sizeof(float), s, &pm); if (rc_ws) return rc_ws; }
      hipLaunchKernelGGL(c8i_to_pixel_major_kernel, dim3((unsigned)cdiv_sz((size_t)g->feat_h * g->feat_w * Cb * 2, 256)), dim3(256), 0, s, g->feat, Cb,
                         (size_t)g->feat_h * g->feat_w, fa.pitch(), static_cast<float *>(pm));
      MPN_CHECK_LAUNCH();
      const Act fdim{nullptr, g->feat_c, g->feat_h, g->feat_w, 0, 0};
      int rcp = roi_pool_pm(fdim, static_cast<const float *>(pm), d_rois, N, PH, PH, spatial_scale, RoiRule{1.0f, 0, g->roi_bins}, g->fc_x, s, roi_stride, round_up(N, 128));
      if (rcp) return rcp;
    } else if ((g_bf16_fast_pool & 1) && Cb % 4 == 0)
      hipLaunchKernelGGL(roi_pool_c8i_rows_kernel<4>, dim3((unsigned)cdiv_sz((size_t)N * PH * PH, 256), (unsigned)(Cb / 4)), dim3(256), 0, s, g->feat, g->feat_h,
                         g->feat_w, fa.pitch(), d_rois, roi_stride, N, PH, PH, spatial_scale, fc_gemm ? g->fc_x : pool_dst, pa.pitch(), fc_gemm ? round_up(N, 128) : 0, g->roi_bins);
    else
      hipLaunchKernelGGL(roi_pool_c8i_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, g->feat, Cb, g->feat_h, g->feat_w, fa.pitch(), d_rois, roi_stride,
                         N, PH, PH, spatial_scale, pool_dst, pa.pitch(), g->roi_bins);
    MPN_CHECK_LAUNCH();
  }
  ActI cur{pool_dst, N, g->feat_c, PH, PH}, y;
  if (g->is_graph) {
    int rc = graph_run(g, g->g_heads[head], t_head, N, s, fuse_mp, fc_gemm);
    if (rc) return rc;
    const GTensor &o = t_head[g->out_tensor];
    cur = ActI{o.buf, N, o.C, o.H, o.W};
  } else
  for (auto &blk : g->heads[head]) {
#ifdef MPN_DEBUG_HOOKS
    if (g_tower_knock && &blk == &g->heads[head][0]) g_knock_arm = blk.has_sc ? 2 : 1;  // block 1's shortcut and conv1 read the pooled tensor
#endif
    int rc = rn_block(blk, cur, hb, s, &y, true);
#ifdef MPN_DEBUG_HOOKS
    g_knock_arm = 0;
#endif
    if (rc) return rc;
    cur = y;
  }
  const size_t total = (size_t)N * cur.Cb() * 2;
  if (g->bf16 && (g_bf16_fast_pool & 2) && (size_t)16 * cur.H * cur.W * 9 * sizeof(float) <= 64 * 1024)
    hipLaunchKernelGGL(avgpool_c8i_bf16_to_c8_lds_kernel, dim3((unsigned)((N + 15) / 16), (unsigned)cur.Cb()), dim3(256), (size_t)16 * cur.H * cur.W * 9 * sizeof(float), s,
                       reinterpret_cast<const bf16_t *>(cur.p), N, cur.H * cur.W, cur.pitch(), 1.0f / (float)(cur.H * cur.W), d_feat_c8, Mp);
  else if (g->bf16)
    hipLaunchKernelGGL(avgpool_c8i_bf16_to_c8_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, reinterpret_cast<const bf16_t *>(cur.p), N, cur.Cb(),
                       cur.H * cur.W, cur.pitch(), 1.0f / (float)(cur.H * cur.W), d_feat_c8, Mp);
  else
    hipLaunchKernelGGL(avgpool_c8i_to_c8_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, cur.p, N, cur.Cb(), cur.H * cur.W, cur.pitch(),
                       1.0f / (float)(cur.H * cur.W), d_feat_c8, Mp);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

}  // namespace mpn

#ifdef MPN_DEBUG_HOOKS
// Kernel-only timing of ONE per-ROI bf16 convolution (rn_conv, per_roi dispatch) on a batch of B maps of H x W (tools/bench_conv_bf16.py):
// random bf16 activations / weights (zero operands clock higher), optional residual; returns the average ms of `iters` launches.
namespace mpn {
__global__ void checksum_u16_kernel(const unsigned short *__restrict__ p, size_t n, unsigned long long *__restrict__ acc) {
  unsigned long long local = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    local += (unsigned long long)p[i] * (unsigned long long)(i % 65521u + 1u);
  atomicAdd(acc, local);
}
}  // namespace mpn
// checksum_out (optional): a position-weighted 64-bit checksum of the (zero-initialised, then written) output tensor — equal checksums
// under two kernel choices = the same bits
extern "C" int mpn_debug_bench_conv_bf16(int Cin, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int B, int H, int W, int with_res, int iters,
                                         float *ms_out, unsigned long long *checksum_out) {
  using namespace mpn;
  MPN_CHECK_ARG(Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && B > 0 && H > 0 && W > 0 && iters > 0 && ms_out);
  ResNetGraph *g = new ResNetGraph();
  g->bf16 = true;
  RnConv c;
  c.Cin = Cin; c.Cout = Cout; c.KH = KH; c.KW = KW; c.sh = sh; c.sw = sw; c.ph = ph; c.pw = pw; c.K = KH; c.stride = sh; c.pad = ph;
  const size_t nw = (size_t)Cout * Cin * KH * KW;
  std::vector<float> hw(nw);
  unsigned x = 2463534242u;
  for (auto &v : hw) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.05f; }
  float *d_w = nullptr;
  int rc = rn_alloc(g, &d_w, nw * sizeof(float));
  if (rc == MPN_OK && hipMemcpy(d_w, hw.data(), nw * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = MPN_EHIP;
  if (rc == MPN_OK) rc = rn_pack(g, c, d_w, nullptr);
  const int OH = (H + 2 * ph - KH) / sh + 1, OW = (W + 2 * pw - KW) / sw + 1;
  const size_t ie = c8i_elems(B, Cin, H, W), oe = c8i_elems(B, Cout, OH, OW);
  float *in = nullptr, *out = nullptr, *res = nullptr;
  if (rc == MPN_OK) rc = rn_alloc(g, &in, ie * sizeof(bf16_t));
  if (rc == MPN_OK) rc = rn_alloc(g, &out, oe * sizeof(bf16_t));
  if (rc == MPN_OK && with_res) rc = rn_alloc(g, &res, oe * sizeof(bf16_t));
  if (rc == MPN_OK) {
    std::vector<unsigned short> hi(std::max(ie, oe));
    for (auto &v : hi) { x = x * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((x >> 9) & 0x3ffu) + ((x >> 31) << 15)); }  // +-[0.0078, 0.031): finite bf16
    { const char *fill = getenv("MPN_BENCH_FILL"); if (fill && fill[0] == 'z') for (auto &v : hi) v = 0; }  // DVFS experiments: zero activations clock higher
    if (hipMemcpy(in, hi.data(), ie * sizeof(bf16_t), hipMemcpyHostToDevice) != hipSuccess) rc = MPN_EHIP;
    if (rc == MPN_OK && res && hipMemcpy(res, hi.data(), oe * sizeof(bf16_t), hipMemcpyHostToDevice) != hipSuccess) rc = MPN_EHIP;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == MPN_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = MPN_EHIP;
  const ActI ai{in, B, Cin, H, W};
  ActI o;
  for (int i = 0; i < 2 && rc == MPN_OK; ++i) rc = rn_conv(c, ai, out, res, 1, nullptr, &o, true, true);
  if (rc == MPN_OK && hipDeviceSynchronize() != hipSuccess) rc = MPN_EHIP;
  if (rc == MPN_OK && checksum_out) {
    unsigned long long *d_acc = nullptr;
    if (hipMalloc(&d_acc, 8) != hipSuccess || hipMemset(d_acc, 0, 8) != hipSuccess || hipMemset(out, 0, oe * sizeof(bf16_t)) != hipSuccess) rc = MPN_EHIP;
    if (rc == MPN_OK) rc = rn_conv(c, ai, out, res, 1, nullptr, &o, true, true);
    if (rc == MPN_OK) {
      hipLaunchKernelGGL(checksum_u16_kernel, dim3(2048), dim3(256), 0, nullptr, reinterpret_cast<const unsigned short *>(out), oe, d_acc);
      if (hipMemcpy(checksum_out, d_acc, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = MPN_EHIP;
    }
    if (d_acc) (void)hipFree(d_acc);
  }
  if (rc == MPN_OK) (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters && rc == MPN_OK; ++i) rc = rn_conv(c, ai, out, res, 1, nullptr, &o, true, true);
  if (rc == MPN_OK) {
    (void)hipEventRecord(e1, nullptr);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = MPN_EHIP;
    *ms_out = ms / iters;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  resnet_free(g);
  return rc;
}
extern "C" void mpn_debug_set_tower_knock(int v) { mpn::g_tower_knock = v; }
extern "C" void mpn_debug_set_pool_exp(int v) { mpn::g_pool_exp = v; }
extern "C" void mpn_debug_set_bf16_dma(int v) { mpn::g_bf16_dma = v; }
extern "C" void mpn_debug_set_roi_invariant(int v) { mpn::g_roi_invariant = v; }
extern "C" void mpn_debug_set_bf16_exp(int v) { mpn::g_bf16_exp = v; }
extern "C" void mpn_debug_set_bf16_bdir(int v) { mpn::g_bf16_bdir = v; }
extern "C" void mpn_debug_set_bf16_bdir_ver(int v) { mpn::g_bf16_bdir_ver = v; }
extern "C" void mpn_debug_set_bf16_bdir_abl(int v) { mpn::g_bf16_bdir_abl = v; }
extern "C" void mpn_debug_set_bf16_nch(int v) { mpn::g_bf16_nch = v; }
extern "C" void mpn_debug_set_bf16_dma_tn(int v) { mpn::g_bf16_dma_tn = v; }
extern "C" void mpn_debug_set_bf16_split_target(int v) { mpn::g_bf16_split_target = v; }
extern "C" void mpn_debug_set_bf16_fast_pool(int v) { mpn::g_bf16_fast_pool = v; }
extern "C" void mpn_debug_set_graph_fuse(int v) { mpn::g_graph_fuse = v; }
extern "C" void mpn_debug_set_fp32_pf(int v) { mpn::g_fp32_pf = v; }
extern "C" void mpn_debug_set_bf16_trace(void *p, int kh) { mpn::g_bf16_trace = static_cast<unsigned long long *>(p); mpn::g_bf16_trace_kh = kh; }
extern "C" void mpn_debug_set_split_max_tiles(int v) { mpn::g_split_max_tiles = v; }
#endif  // MPN_DEBUG_HOOKS
