// nms.hip — wavefront NMS / bbox voting for gfx950.  Replaces nms.c (reference's only native code).
//
// Design (MI355X-first, not a translation of the serial C loop):
//   * one 64-lane wavefront owns one class; the class's boxes live in LDS as SoA (x1,y1,x2,y2,s,pos),
//     so consecutive lanes read consecutive banks (conflict-free ds_read_b32);
//   * a greedy round = one fused sweep: every lane tests its surviving boxes against the box picked in
//     the previous round (IoU, nms.c:14-41), drops the suppressed ones and, in the same pass, finds its
//     local candidate for the next pick and the "first alive" element; two butterfly reductions over
//     the wave (DPP/ds_swizzle via __shfl_xor) finish the round.  No block barrier, no global traffic;
//   * the reference's winner among bit-equal scores depends on its swap + stable-partition history
//     (nms.c:74-98).  It is reproduced exactly with a per-box position key `pos`: the pick is the
//     alive box maximising (score, -pos); the element that sat first in the array (min pos) inherits
//     the picked box's pos (the nms.c:83-85 swap); the stable partition keeps every other relative
//     order, so no other key changes.
//   * classes are independent -> grid = n_cls blocks of one wave; the per-image latency is the longest
//     class, the launch fills n_cls of the 256 CUs and is meant to overlap the next image's trunk.
#include "mpn_internal.h"

namespace mpn {

struct Cand {
  float s;   // score
  int pos;   // position key (smaller = earlier in the reference's array)
  int idx;   // original row index, -1 = none
};

__device__ __forceinline__ bool better(float s, int pos, const Cand &c) {
  // strict '>' on the score picks the FIRST maximum in array order (nms.c:77-80)
  return c.idx < 0 || s > c.s || (s == c.s && pos < c.pos);
}

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    Cand o;
    o.s = __shfl_xor(c.s, off);
    o.pos = __shfl_xor(c.pos, off);
    o.idx = __shfl_xor(c.idx, off);
    if (o.idx >= 0 && better(o.s, o.pos, c)) c = o;
  }
  return c;
}

// min over (pos, idx) pairs; idx<0 = none
__device__ __forceinline__ void wave_first(int &pos, int &idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    int op = __shfl_xor(pos, off);
    int oi = __shfl_xor(idx, off);
    if (oi >= 0 && (idx < 0 || op < pos)) { pos = op; idx = oi; }
  }
}

constexpr int kDead = 0x7fffffff;

// LDS: 6 arrays of m_cap entries (x1,y1,x2,y2,score as float; pos as int; pos==kDead marks removed)
__global__ __launch_bounds__(64) void nms_wave_kernel(const float *__restrict__ scored, const int *__restrict__ counts,
                                                      int m_stride, float thr, float *__restrict__ keep,
                                                      int *__restrict__ keep_idx, int *__restrict__ n_keep,
                                                      int m_cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *X1 = lds, *Y1 = lds + m_cap, *X2 = lds + 2 * m_cap, *Y2 = lds + 3 * m_cap, *S = lds + 4 * m_cap;
  int *POS = reinterpret_cast<int *>(lds + 5 * m_cap);

  const int cls = blockIdx.x;
  const int lane = threadIdx.x;
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  const float *src = scored + (size_t)cls * m_stride * 5;
  float *kout = keep + (size_t)cls * m_stride * 5;
  int *kidx = keep_idx ? keep_idx + (size_t)cls * m_stride : nullptr;

  // stage (coalesced over the flat [m,5] array), then every lane scans for the first pick
  for (int t = lane; t < m * 5; t += kWave) {
    float v = src[t];
    int i = t / 5, f = t - 5 * i;
    (f == 0 ? X1 : f == 1 ? Y1 : f == 2 ? X2 : f == 3 ? Y2 : S)[i] = v;
  }
  for (int i = lane; i < m; i += kWave) POS[i] = i;
  __syncthreads();

  Cand best{0.f, 0, -1};
  int fpos = 0, fidx = -1;
  for (int i = lane; i < m; i += kWave) {
    float s = S[i];
    if (s > -10000000.0f && better(s, i, best)) best = Cand{s, i, i};  // nms.c:75 bestS init
    if (fidx < 0) { fpos = i; fidx = i; }
  }
  best = wave_best(best);
  wave_first(fpos, fidx);

  int kept = 0;
  while (best.idx >= 0) {
    const int b = best.idx;
    // nms.c:83-85: boxes[0] <-> boxes[best]; the old first element now sits where `best` sat
    if (lane == 0) {
      if (fidx != b) POS[fidx] = best.pos;
      POS[b] = kDead;
    }
    const float bx1 = X1[b], by1 = Y1[b], bx2 = X2[b], by2 = Y2[b];
    if (lane < 5) kout[(size_t)kept * 5 + lane] = lane == 0 ? bx1 : lane == 1 ? by1 : lane == 2 ? bx2 : lane == 3 ? by2 : best.s;
    if (lane == 0 && kidx) kidx[kept] = b;
    ++kept;
    __syncthreads();  // single wave: orders lane 0's POS writes before the sweep

    Cand nb{0.f, 0, -1};
    int nfpos = 0, nfidx = -1;
    for (int i = lane; i < m; i += kWave) {
      int p = POS[i];
      if (p == kDead) continue;
      float iou = iou_plus1(bx1, by1, bx2, by2, X1[i], Y1[i], X2[i], Y2[i]);
      if (!(iou <= thr)) {  // nms.c:93 keeps `iou <= threshold`; NaN is dropped like the reference
        POS[i] = kDead;
        continue;
      }
      float s = S[i];
      if (s > -10000000.0f && better(s, p, nb)) nb = Cand{s, p, i};
      if (nfidx < 0 || p < nfpos) { nfpos = p; nfidx = i; }
    }
    best = wave_best(nb);
    fpos = nfpos; fidx = nfidx;
    wave_first(fpos, fidx);
    __syncthreads();
  }
  if (lane == 0) n_keep[cls] = kept;
}

// nms.c:110-142.  One wave per kept box would reorder the sequential fp32 sums, so each LANE owns one
// kept box and walks the scored boxes (staged in LDS tiles) in j order: the accumulation order — and
// hence every rounding — is the reference's.
__global__ __launch_bounds__(64) void bbox_vote_kernel(const float *__restrict__ nmsb, int n_nms_arg,
                                                       const int *__restrict__ d_n_nms,
                                                       const float *__restrict__ scored, int m, float thr,
                                                       float *__restrict__ res) {
  constexpr int TILE = 256;
  __shared__ float t[TILE * 5];
  const int n_nms = d_n_nms ? min(*d_n_nms, n_nms_arg) : n_nms_arg;
  const int i = blockIdx.x * kWave + threadIdx.x;
  const bool act = i < n_nms;
  float nx1 = 0, ny1 = 0, nx2 = 0, ny2 = 0, ns = 0;
  if (act) { nx1 = nmsb[5 * i]; ny1 = nmsb[5 * i + 1]; nx2 = nmsb[5 * i + 2]; ny2 = nmsb[5 * i + 3]; ns = nmsb[5 * i + 4]; }
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int j0 = 0; j0 < m; j0 += TILE) {
    int cnt = min(TILE, m - j0);
    __syncthreads();
    for (int q = threadIdx.x; q < cnt * 5; q += kWave) t[q] = scored[(size_t)j0 * 5 + q];
    __syncthreads();
    if (act) {
      for (int j = 0; j < cnt; ++j) {
        float sx1 = t[5 * j], sy1 = t[5 * j + 1], sx2 = t[5 * j + 2], sy2 = t[5 * j + 3], ss = t[5 * j + 4];
        float ov = iou_plus1(sx1, sy1, sx2, sy2, nx1, ny1, nx2, ny2);  // overlap(scored_j, nms_i)
        if (ov > thr) {
          a0 += sx1 * ss; a1 += sy1 * ss; a2 += sx2 * ss; a3 += sy2 * ss; a4 += ss;
        }
      }
    }
  }
  if (act) {
    res[5 * i] = a0 / a4; res[5 * i + 1] = a1 / a4; res[5 * i + 2] = a2 / a4; res[5 * i + 3] = a3 / a4;
    res[5 * i + 4] = ns;
  }
}

__global__ void boxoverlap_kernel(const float *__restrict__ a, int n, float bx1, float by1, float bx2, float by2,
                                  float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = iou_plus1(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3], bx1, by1, bx2, by2);
}

}  // namespace mpn

using namespace mpn;

extern "C" int mpn_nms_batched(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr,
                               float *d_keep, int *d_keep_idx, int *d_n_keep, void *stream) {
  MPN_CHECK_ARG(n_cls >= 0 && m_stride >= 0);
  MPN_CHECK_ARG(m_stride <= MPN_NMS_MAX_BOXES);
  MPN_CHECK_ARG(d_n_keep != nullptr);
  if (n_cls == 0) return MPN_OK;
  if (m_stride == 0) {
    MPN_CHECK_HIP(hipMemsetAsync(d_n_keep, 0, sizeof(int) * n_cls, as_stream(stream)));
    return MPN_OK;
  }
  MPN_CHECK_ARG(d_scored != nullptr && d_keep != nullptr);
  int m_cap = (m_stride + 3) & ~3;
  size_t lds = (size_t)m_cap * 6 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(nms_wave_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, MPN_NMS_MAX_BOXES * 6 * 4));
    attr_set = true;
  }
  hipLaunchKernelGGL(nms_wave_kernel, dim3(n_cls), dim3(kWave), lds, as_stream(stream), d_scored, d_counts, m_stride,
                     thr, d_keep, d_keep_idx, d_n_keep, m_cap);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_nms(const float *d_scored, int m, float thr, float *d_keep, int *d_keep_idx, int *d_n_keep,
                       void *stream) {
  return mpn_nms_batched(d_scored, nullptr, 1, m, thr, d_keep, d_keep_idx, d_n_keep, stream);
}

extern "C" int mpn_nms_host(const float *h_scored, int m, float thr, float *h_keep, int *h_keep_idx, int *n_keep) {
  MPN_CHECK_ARG(m >= 0 && n_keep != nullptr);
  *n_keep = 0;
  if (m == 0) return MPN_OK;
  MPN_CHECK_ARG(h_scored != nullptr && h_keep != nullptr);
  float *d_in = nullptr, *d_keep = nullptr;
  int *d_idx = nullptr, *d_n = nullptr;
  size_t bytes = sizeof(float) * 5 * (size_t)m;
  MPN_CHECK_HIP(hipMalloc(&d_in, bytes * 2 + sizeof(int) * ((size_t)m + 1)));
  d_keep = d_in + 5 * (size_t)m;
  d_idx = reinterpret_cast<int *>(d_keep + 5 * (size_t)m);
  d_n = d_idx + m;
  int rc = MPN_OK;
  hipError_t e = hipMemcpy(d_in, h_scored, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = mpn_nms(d_in, m, thr, d_keep, d_idx, d_n, nullptr);
    if (rc == MPN_OK) e = hipMemcpy(n_keep, d_n, sizeof(int), hipMemcpyDeviceToHost);
    if (rc == MPN_OK && e == hipSuccess && *n_keep > 0) {
      e = hipMemcpy(h_keep, d_keep, sizeof(float) * 5 * (size_t)*n_keep, hipMemcpyDeviceToHost);
      if (e == hipSuccess && h_keep_idx) e = hipMemcpy(h_keep_idx, d_idx, sizeof(int) * (size_t)*n_keep, hipMemcpyDeviceToHost);
    }
  }
  (void)hipFree(d_in);
  if (rc != MPN_OK) return rc;
  if (e != hipSuccess) { set_error("mpn_nms_host: %s", hipGetErrorString(e)); return MPN_EHIP; }
  return MPN_OK;
}

extern "C" int mpn_bbox_vote(const float *d_nms, int n_nms, const int *d_n_nms, const float *d_scored, int m,
                             float thr, float *d_res, void *stream) {
  MPN_CHECK_ARG(n_nms >= 0 && m >= 0);
  if (n_nms == 0) return MPN_OK;
  MPN_CHECK_ARG(d_nms != nullptr && d_res != nullptr && (m == 0 || d_scored != nullptr));
  hipLaunchKernelGGL(bbox_vote_kernel, dim3(cdiv(n_nms, kWave)), dim3(kWave), 0, as_stream(stream), d_nms, n_nms,
                     d_n_nms, d_scored, m, thr, d_res);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_bbox_vote_host(const float *h_nms, int n_nms, const float *h_scored, int m, float thr,
                                  float *h_res) {
  MPN_CHECK_ARG(n_nms >= 0 && m >= 0);
  if (n_nms == 0) return MPN_OK;
  MPN_CHECK_ARG(h_nms && h_res && (m == 0 || h_scored));
  float *d = nullptr;
  size_t nb = sizeof(float) * 5 * (size_t)n_nms, sb = sizeof(float) * 5 * (size_t)m;
  MPN_CHECK_HIP(hipMalloc(&d, 2 * nb + sb + 16));
  float *d_nms = d, *d_res = d + 5 * (size_t)n_nms, *d_sc = d_res + 5 * (size_t)n_nms;
  hipError_t e = hipMemcpy(d_nms, h_nms, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess && m > 0) e = hipMemcpy(d_sc, h_scored, sb, hipMemcpyHostToDevice);
  int rc = MPN_OK;
  if (e == hipSuccess) {
    rc = mpn_bbox_vote(d_nms, n_nms, nullptr, d_sc, m, thr, d_res, nullptr);
    if (rc == MPN_OK) e = hipMemcpy(h_res, d_res, nb, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (rc != MPN_OK) return rc;
  if (e != hipSuccess) { set_error("mpn_bbox_vote_host: %s", hipGetErrorString(e)); return MPN_EHIP; }
  return MPN_OK;
}

extern "C" int mpn_boxoverlap(const float *d_a, int n, const float *h_b, float *d_out, void *stream) {
  MPN_CHECK_ARG(n >= 0 && h_b != nullptr);
  if (n == 0) return MPN_OK;
  MPN_CHECK_ARG(d_a != nullptr && d_out != nullptr);
  hipLaunchKernelGGL(boxoverlap_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), d_a, n, h_b[0], h_b[1],
                     h_b[2], h_b[3], d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}
