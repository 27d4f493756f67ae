// boxes.hip — the small HBM/latency-bound kernels around the dense path: image transformer,
// ROI projection, Foveal / ContextRegion / BBoxNorm / SelectBoxes modules, softmax, bbox decode,
// clamp, per-class scored-box selection and the global top-k.  Each kernel cites the reference
// lines it mirrors; fp32 arithmetic follows the reference's operation order (no FMA contraction).
#include "mpn_internal.h"

namespace mpn {

// modules/ImageTransformer.lua:19-33 — f64 arithmetic, one rounding to fp32 (see mpn.h).
__global__ void image_transform_kernel(const float *__restrict__ in, size_t plane, int s0, int s1, int s2, double scale,
                                       double m0, double m1, double m2, double d0, double d1, double d2, int has_std,
                                       float *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = blockIdx.y;
  if (i >= plane) return;
  int sc = c == 0 ? s0 : c == 1 ? s1 : s2;
  double mean = c == 0 ? m0 : c == 1 ? m1 : m2;
  double sd = c == 0 ? d0 : c == 1 ? d1 : d2;
  double v = (double)in[(size_t)sc * plane + i];
  if (scale != 1.0) v = v * scale;
  v = v + (-mean);
  if (has_std) v = v / sd;
  out[(size_t)c * plane + i] = (float)v;
}

// ImageDetect.lua:66-70
__global__ void project_rois_kernel(const float *__restrict__ boxes, int n, float s, float *__restrict__ rois, float *__restrict__ boxes_copy) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rois[5 * i] = 1.0f;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float v = boxes[4 * i + f];
    if (boxes_copy) boxes_copy[4 * i + f] = v;  // the pipelined forms keep the caller's boxes per buffer set (deferred decode)
    v = v + (-1.0f);
    v = v * s;
    v = v + 1.0f;
    rois[5 * i + 1 + f] = v;
  }
}

// modules/Foveal.lua:15-44 — Lua doubles, rounded to fp32 on store.
__global__ void foveal_kernel(const float *__restrict__ rois, int n, float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *r = rois + 5 * (size_t)i;
  float fid = r[0], fx = r[1], fy = r[2], fx2 = r[3], fy2 = r[4];
  double id = fid, x = fx, y = fy, x2 = fx2, y2 = fy2;
  double w = x2 - x, h = y2 - y;
  float *o = out + 20 * (size_t)i;
  o[0] = fid; o[1] = fx; o[2] = fy; o[3] = fx2; o[4] = fy2;
  const double off[3] = {0.25, 0.5, 1.5};
  const double mul[3] = {1.5, 2.0, 4.0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double xx = x - w * off[k], yy = y - h * off[k], ww = w * mul[k], hh = h * mul[k];
    o[5 * (k + 1) + 0] = (float)id;
    o[5 * (k + 1) + 1] = (float)xx;
    o[5 * (k + 1) + 2] = (float)yy;
    o[5 * (k + 1) + 3] = (float)(xx + ww);
    o[5 * (k + 1) + 4] = (float)(yy + hh);
  }
}

// modules/ContextRegion.lua:14-32
__global__ void context_region_kernel(const float *__restrict__ rois, int n, float a, float b, float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *r = rois + 5 * (size_t)i;
  float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
  float *o = out + 5 * (size_t)i;
  o[0] = r0;
  o[1] = a * r1 + b * r3;
  o[2] = a * r2 + b * r4;
  o[3] = b * r1 + a * r3;
  o[4] = b * r2 + a * r4;
}

// modules/BBoxNorm.lua:28-29
__global__ void bbox_norm_kernel(float *__restrict__ bbox, size_t total, float m0, float m1, float m2, float m3,
                                 float s0, float s1, float s2, float s3) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int f = (int)(i & 3);
  float sd = f == 0 ? s0 : f == 1 ? s1 : f == 2 ? s2 : s3;
  float mn = f == 0 ? m0 : f == 1 ? m1 : f == 2 ? m2 : m3;
  float v = bbox[i] * sd;
  bbox[i] = v + mn;
}

// modules/SelectBoxes.lua:26-56 (torch.max over dim 2 returns the first maximum)
__global__ void select_boxes_kernel(const float *__restrict__ scores, const float *__restrict__ bbox, int n, int C,
                                    float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *s = scores + (size_t)i * C;
  int best = 0;
  float bs = s[0];
  for (int c = 1; c < C; ++c) {
    float v = s[c];
    if (v > bs) { bs = v; best = c; }
  }
  const float *b = bbox + (size_t)i * 4 * C + 4 * best;
  float *o = out + 4 * (size_t)i;
  o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
}

// nn.SoftMax over the class dim: one wave per row, shuffle reductions; exp(x-max)/sum.
// The sum is accumulated across lanes (tree), so it matches a serial CPU softmax to ~1 ulp, not bit-exactly.
__global__ __launch_bounds__(256) void softmax_kernel(const float *__restrict__ x, int M, int C, float *__restrict__ y) {
  int row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  int lane = threadIdx.x & (kWave - 1);
  if (row >= M) return;
  const float *r = x + (size_t)row * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += kWave) mx = fmaxf(mx, r[c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int c = lane; c < C; c += kWave) sum += expf(r[c] - mx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  for (int c = lane; c < C; c += kWave) y[(size_t)row * C + c] = expf(r[c] - mx) / sum;
}

// utils.lua:229-247 — one thread per (roi, class) 4-vector.
__global__ void bbox_decode_kernel(const float *__restrict__ boxes, const float *__restrict__ deltas, int n, int C,
                                   float *__restrict__ out, int clamp, float im_w, float im_h) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * C) return;
  int i = (int)(t / C);
  const float4 bx = *reinterpret_cast<const float4 *>(boxes + 4 * (size_t)i);
  const float4 d = *reinterpret_cast<const float4 *>(deltas + 4 * t);
  float xc = (bx.x + bx.z) * 0.5f, yc = (bx.y + bx.w) * 0.5f;
  float w = bx.z - bx.x, h = bx.w - bx.y;
  float p0 = d.x * w, p1 = d.y * h;
  float xtc = xc + p0, ytc = yc + p1;
  float wt = expf(d.z) * w, ht = expf(d.w) * h;
  float hw = wt * 0.5f, hh = ht * 0.5f;
  float4 o;
  o.x = xtc - hw; o.y = ytc - hh; o.z = xtc + hw; o.w = ytc + hh;
  if (clamp) {  // Tester_FRCNN.lua:75-78 fused
    o.x = o.x < 1.0f ? 1.0f : (o.x > im_w ? im_w : o.x);
    o.z = o.z < 1.0f ? 1.0f : (o.z > im_w ? im_w : o.z);
    o.y = o.y < 1.0f ? 1.0f : (o.y > im_h ? im_h : o.y);
    o.w = o.w < 1.0f ? 1.0f : (o.w > im_h ? im_h : o.w);
  }
  *reinterpret_cast<float4 *>(out + 4 * t) = o;
}

__global__ void clamp_kernel(float *__restrict__ bbox, size_t n_pairs, float im_w, float im_h) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  float2 v = *reinterpret_cast<float2 *>(bbox + 2 * i);
  v.x = v.x < 1.0f ? 1.0f : (v.x > im_w ? im_w : v.x);
  v.y = v.y < 1.0f ? 1.0f : (v.y > im_h ? im_h : v.y);
  *reinterpret_cast<float2 *>(bbox + 2 * i) = v;
}

// Tester_FRCNN.lua:106-116 for every foreground class at once.  One block per class; rows are
// compacted IN ROW ORDER (the order utils.nms then sees) with a ballot/popcount prefix per wave and
// a running block offset, so the kept order is deterministic and equals the reference's.
__global__ __launch_bounds__(256) void select_scored_kernel(const float *__restrict__ scores,
                                                            const float *__restrict__ bbox, int n, int C,
                                                            int first_cls, float thresh, float *__restrict__ scored,
                                                            int *__restrict__ counts, int *__restrict__ src_idx) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int cls = first_cls + blockIdx.x;
  float *dst = scored + (size_t)blockIdx.x * n * 5;
  int *didx = src_idx ? src_idx + (size_t)blockIdx.x * n : nullptr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    int i = i0 + threadIdx.x;
    float s = 0.f;
    bool take = false;
    if (i < n) { s = scores[(size_t)i * C + cls]; take = s > thresh; }
    unsigned long long mask = __ballot(take);
    int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wid] = __popcll(mask);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wid; ++w) off += wave_cnt[w];
    if (take) {
      int o = off + before;
      const float4 b = *reinterpret_cast<const float4 *>(bbox + (size_t)i * 4 * C + 4 * cls);
      float *p = dst + 5 * (size_t)o;
      p[0] = b.x; p[1] = b.y; p[2] = b.z; p[3] = b.w; p[4] = s;
      if (didx) didx[o] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = base_s;
}

// utils.lua:75-96.  Single block.  The k-th largest score is found by an MSB-first radix select over
// the order-preserving integer image of the fp32 scores (4 passes x 8 bits, LDS histogram) — exact,
// no sort, no host round trip.  Then survivors (score >= thresh) are compacted class-major, in NMS
// selection order inside a class (= the order keep_top_k preserves).
__device__ __forceinline__ unsigned f2key(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;  // -0.0f == +0.0f in utils.keep_top_k's `score >= thresh` (and in nms.c's order): one key for both
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// Layout of the work: the kept rows of all classes form ONE flattened list (class-major, NMS selection order inside a
// class = the order utils.keep_top_k preserves).  Their order-preserving integer keys are staged once in LDS (up to
// kTopkLdsKeys of them; beyond that the passes re-read HBM), so the 4 radix passes and the compaction touch HBM once.
// Wave w owns the contiguous slice [w*seg, (w+1)*seg) of the list: its survivors are counted and later written at
// (sum of earlier waves' counts) + rank-in-slice — an ordered compaction with two block barriers in total.
constexpr int kTopkLdsKeys = 32768;  // 128 KiB of the CU's 160 KiB
constexpr int kTopkMaxCls = 1024;
constexpr int kTopkCand = 2048;   // keep_top_k: candidate list of the radix select (keys of the bin the first pass picked)

__device__ __forceinline__ int topk_class_of(const int *__restrict__ cls_off, int n_cls, int i) {  // last c with cls_off[c] <= i
  int lo = 0, hi = n_cls - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cls_off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

#ifdef MPN_DEBUG_HOOKS
__device__ unsigned long long g_topk_trace[8];  // s_memtime stamps of thread 0 at the phase boundaries (tools/topk_trace.py)
#define TOPK_STAMP(i) do { if (threadIdx.x == 0) g_topk_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TOPK_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(1024) void keep_top_k_kernel(const float *__restrict__ keep, const int *__restrict__ n_keep,
                                                          int n_cls, int m_stride, int k, float *__restrict__ thresh_out,
                                                          float *__restrict__ out, int max_out, int *__restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned topk_keys[];  // [min(total, kTopkLdsKeys)]
  __shared__ int cls_off[kTopkMaxCls + 1];
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_bin, sel_rank, sel_cnt;
  __shared__ int cand_n;
  __shared__ unsigned wave_lo[16], wave_hi[16];
  __shared__ int wave_cnt[16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  TOPK_STAMP(0);
  // ---- class offsets (exclusive prefix sum of the per-class counts): one wave, 64 classes per step
  if (wid == 0) {
    int run = 0;
    for (int c0 = 0; c0 < n_cls; c0 += 64) {
      const int c = c0 + lane;
      int v = c < n_cls ? min(n_keep[c], m_stride) : 0;
      if (v < 0) v = 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      if (c < n_cls) cls_off[c] = run + incl - v;
      run += __shfl(incl, 63);
    }
    if (lane == 0) cls_off[n_cls] = run;
  }
  __syncthreads();
  const int total = cls_off[n_cls];
  if (total == 0) {
    if (tid == 0) { *thresh_out = 0.0f; *n_out = 0; }  // utils.lua:77-79
    return;
  }
  const int n_lds = min(total, kTopkLdsKeys);
  TOPK_STAMP(1);
  auto load_key = [&](int i) -> unsigned {
    const int c = topk_class_of(cls_off, n_cls, i);
    return f2key(keep[((size_t)c * m_stride + (i - cls_off[c])) * 5 + 4]);
  };
  // key staging: 8 independent gathers in flight per thread (a one-load-per-trip loop pays the ~1.5 us global round trip 13
  // times in a row at 13 k detections: most of the kernel's time)
  unsigned mn = 0xffffffffu, mx = 0u;  // the keys' range, gathered while they pass through registers
  for (int base = tid; base < n_lds; base += 8 * (int)blockDim.x) {
    unsigned kv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x;
      kv[u] = i < n_lds ? load_key(i) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x;
      if (i < n_lds) { topk_keys[i] = kv[u]; mn = min(mn, kv[u]); mx = max(mx, kv[u]); }
    }
  }
  TOPK_STAMP(2);
  if (tid == 0) sel_rank = (unsigned)max(min(total, k), 1);  // 1-based rank from the top
  // ---- radix select of the k-th largest key on digits of the keys' OWN range: scores of one image share their top bits (the
  // top byte of 13 k detection scores takes 2-3 values), and a histogram on raw bytes serialises ~13 k LDS atomics on those
  // bins (measured: 20 us of the kernel's 28).  Each pass histograms (key - lo) >> shift over the current [lo, hi] with
  // shift = bits(hi - lo) - 8, picks the bin that holds the rank and narrows [lo, hi] to it: <= 4 passes, ~128 bins in use.
  for (int i = n_lds + tid; i < total; i += blockDim.x) {  // keys beyond the LDS stage (> 32 768 rows): straight from memory
    const unsigned key = load_key(i);
    mn = min(mn, key); mx = max(mx, key);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = min(mn, (unsigned)__shfl_xor((int)mn, off));
    mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
  }
  if (lane == 0) { wave_lo[wid] = mn; wave_hi[wid] = mx; }
  __syncthreads();
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int w = 0; w < nw; ++w) { lo = min(lo, wave_lo[w]); hi = max(hi, wave_hi[w]); }
  TOPK_STAMP(3);
  // After the first pass the rank lives in ONE bin (~1 % of the keys): those keys are compacted into a small LDS list and the
  // remaining passes walk only that list instead of all the keys again (unless ties put more than kTopkCand keys into the bin).
  unsigned *const cand = topk_keys + n_lds + 4;
  bool use_cand = false;
  int ncand = 0;
  if (tid == 0) cand_n = 0;
  while (hi > lo) {  // block-uniform: lo / hi derive from shared values only
    const unsigned width = hi - lo;
    const int nb = 32 - __clz((int)width);
    const int shift = nb > 8 ? nb - 8 : 0;
    for (int b = tid; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    if (use_cand) {
      for (int i = tid; i < ncand; i += blockDim.x) {
        const unsigned key = cand[i];
        if (key >= lo && key <= hi) atomicAdd(&hist[(key - lo) >> shift], 1u);
      }
    } else {
      for (int i = tid; i < total; i += blockDim.x) {
        const unsigned key = i < n_lds ? topk_keys[i] : load_key(i);
        if (key >= lo && key <= hi) atomicAdd(&hist[(key - lo) >> shift], 1u);
      }
    }
    __syncthreads();
    if (wid == 0) {  // the bin that holds the rank: suffix sums over the 256 bins, 4 bins per lane (lane 0 = bins 252..255)
      const int b0 = 252 - 4 * lane;
      const unsigned h3 = hist[b0 + 3], h2 = hist[b0 + 2], h1 = hist[b0 + 1], h0 = hist[b0];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const unsigned before = incl - mine;  // keys in bins above this lane's four
      const unsigned rank = sel_rank;
      const bool hit = before < rank && incl >= rank;
      const unsigned long long hm = __ballot(hit);
      // rank <= (keys inside [lo, hi]) by construction, so exactly one lane hits
      if (hm && lane == __builtin_ctzll(hm)) {
        unsigned acc = before, cnt = h3;
        int b = b0 + 3;
        if (acc + h3 < rank) { acc += h3; b = b0 + 2; cnt = h2; if (acc + h2 < rank) { acc += h2; b = b0 + 1; cnt = h1; if (acc + h1 < rank) { acc += h1; b = b0; cnt = h0; } } }
        sel_rank = rank - acc;
        sel_bin = (unsigned)b;
        sel_cnt = cnt;
      }
    }
    __syncthreads();
    lo += sel_bin << shift;
    hi = min(hi, lo + ((1u << shift) - 1u));
    // (the next pass's hist zeroing is ordered after every thread's read of sel_bin by the barrier that follows it)
    if (!use_cand && hi > lo && sel_cnt <= (unsigned)kTopkCand) {
      for (int i = tid; i < total; i += blockDim.x) {
        const unsigned key = i < n_lds ? topk_keys[i] : load_key(i);
        if (key >= lo && key <= hi) cand[atomicAdd(&cand_n, 1)] = key;
      }
      __syncthreads();
      ncand = cand_n;
      use_cand = true;
    }
  }
  const unsigned thr_key = lo;
  TOPK_STAMP(4);
  if (tid == 0) *thresh_out = key2f(thr_key);
  // ---- ordered compaction of the survivors (key >= threshold key <=> score >= threshold)
  const int seg = (total + nw - 1) / nw;
  const int s0 = min(wid * seg, total), s1 = min(s0 + seg, total);
  int cnt = 0;
  for (int i0 = s0; i0 < s1; i0 += 64) {
    const int i = i0 + lane;
    const bool take = i < s1 && (i < n_lds ? topk_keys[i] : load_key(i)) >= thr_key;
    cnt += __popcll(__ballot(take));
  }
  if (lane == 0) wave_cnt[wid] = cnt;
  __syncthreads();
  TOPK_STAMP(5);
  int off = 0, all = 0;
  for (int w = 0; w < nw; ++w) { const int v = wave_cnt[w]; if (w < wid) off += v; all += v; }
  for (int i0 = s0; i0 < s1; i0 += 64) {
    const int i = i0 + lane;
    const bool take = i < s1 && (i < n_lds ? topk_keys[i] : load_key(i)) >= thr_key;
    const unsigned long long m = __ballot(take);
    const int o = off + __popcll(m & ((1ull << lane) - 1ull));
    if (take && o < max_out) {
      const int c = topk_class_of(cls_off, n_cls, i);
      const float *row = keep + ((size_t)c * m_stride + (i - cls_off[c])) * 5;
      float *q = out + 6 * (size_t)o;
      q[0] = row[0]; q[1] = row[1]; q[2] = row[2]; q[3] = row[3]; q[4] = row[4]; q[5] = (float)(c + 1);
    }
    off += __popcll(m);
  }
  TOPK_STAMP(6);
  if (tid == 0) *n_out = all;  // the UNTRUNCATED survivor count: > max_out tells the caller rows were dropped
}

// keep_top_k on tables whose rows are in NON-INCREASING score order inside every class — what NMS emits (each round picks the largest
// remaining score) and what bbox_vote keeps (nms.c:139: the voted rows keep their NMS scores).  The k-th largest score over all classes is
// then the k-th largest among the FIRST min(k, n_c) rows of each class, and the survivors of a class are a prefix of it: the kernel
// stages n_cls x k keys (8 000 for 80 classes, whatever the tables' height — the general kernel above stages every kept row: 100 k keys
// at 80 x 2000, three quarters of them re-read from HBM by every radix pass: 184 us on BASELINE configs[4]) and never touches the rest,
// except for a class whose whole staged prefix survives (ties at the threshold run past row k: its tail is walked 64 rows at a time).
__global__ __launch_bounds__(1024) void keep_top_k_sorted_kernel(const float *__restrict__ keep, const int *__restrict__ n_keep, int n_cls, int m_stride,
                                                                 int k, float *__restrict__ thresh_out, float *__restrict__ out, int max_out,
                                                                 int *__restrict__ n_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned topk_keys[];  // [n_cls][kk]: slot (c, j) valid iff j < cls_n[c]
  __shared__ int cls_n[kTopkMaxCls];
  __shared__ int cls_off[kTopkMaxCls + 1];  // survivors per class, then their exclusive prefix sum
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_bin, sel_rank;
  __shared__ unsigned wave_lo[16], wave_hi[16];
  __shared__ int total_rows;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  const int kk = min(k, m_stride);
  if (tid == 0) total_rows = 0;
  __syncthreads();
  {
    int part = 0;
    for (int c = tid; c < n_cls; c += blockDim.x) {
      int v = min(n_keep[c], m_stride);
      if (v < 0) v = 0;
      cls_n[c] = v;
      part += v;
    }
    if (part) atomicAdd(&total_rows, part);
  }
  __syncthreads();
  const int total = total_rows;
  if (total == 0) {
    if (tid == 0) { *thresh_out = 0.0f; *n_out = 0; }  // utils.lua:77-79
    return;
  }
  const int n_slot = n_cls * kk;
  unsigned mn = 0xffffffffu, mx = 0u;
  for (int base = tid; base < n_slot; base += 8 * (int)blockDim.x) {  // 8 independent gathers in flight per thread
    unsigned kv[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x;
      const int c = i < n_slot ? i / kk : 0, j = i - c * kk;
      ok[u] = i < n_slot && j < cls_n[c];
      kv[u] = ok[u] ? f2key(keep[((size_t)c * m_stride + j) * 5 + 4]) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * (int)blockDim.x;
      if (i < n_slot) topk_keys[i] = kv[u];
      if (ok[u]) { mn = min(mn, kv[u]); mx = max(mx, kv[u]); }
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = min(mn, (unsigned)__shfl_xor((int)mn, off));
    mx = max(mx, (unsigned)__shfl_xor((int)mx, off));
  }
  if (lane == 0) { wave_lo[wid] = mn; wave_hi[wid] = mx; }
  if (tid == 0) sel_rank = (unsigned)max(min(total, k), 1);
  __syncthreads();
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int w = 0; w < nw; ++w) { lo = min(lo, wave_lo[w]); hi = max(hi, wave_hi[w]); }
  // radix select on digits of the keys' own range (as keep_top_k_kernel)
  while (hi > lo) {
    const unsigned width = hi - lo;
    const int nb = 32 - __clz((int)width);
    const int shift = nb > 8 ? nb - 8 : 0;
    for (int b = tid; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n_slot; i += blockDim.x) {
      const int c = i / kk, j = i - c * kk;
      const unsigned key = topk_keys[i];
      if (j < cls_n[c] && key >= lo && key <= hi) atomicAdd(&hist[(key - lo) >> shift], 1u);
    }
    __syncthreads();
    if (wid == 0) {  // the bin that holds the rank: suffix sums over the 256 bins, 4 bins per lane (lane 0 = bins 252..255)
      const int b0 = 252 - 4 * lane;
      const unsigned h3 = hist[b0 + 3], h2 = hist[b0 + 2], h1 = hist[b0 + 1], h0 = hist[b0];
      const unsigned mine = h0 + h1 + h2 + h3;
      unsigned incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const unsigned before = incl - mine;
      const unsigned rank = sel_rank;
      const bool hit = before < rank && incl >= rank;
      const unsigned long long hm = __ballot(hit);
      if (hm && lane == __builtin_ctzll(hm)) {
        unsigned acc = before;
        int b = b0 + 3;
        if (acc + h3 < rank) { acc += h3; b = b0 + 2; if (acc + h2 < rank) { acc += h2; b = b0 + 1; if (acc + h1 < rank) { acc += h1; b = b0; } } }
        sel_rank = rank - acc;
        sel_bin = (unsigned)b;
      }
    }
    __syncthreads();
    lo += sel_bin << shift;
    hi = min(hi, lo + ((1u << shift) - 1u));
    __syncthreads();  // every thread has read sel_bin before the next pass rewrites it
  }
  const unsigned thr_key = lo;
  if (tid == 0) *thresh_out = key2f(thr_key);
  // survivors of class c = a prefix: counted in the staged keys, continued in the table only when the whole staged prefix survives
  for (int c = wid; c < n_cls; c += nw) {
    const int nc = cls_n[c], staged = min(kk, nc);
    int len = 0;
    for (int j0 = 0; j0 < staged; j0 += 64) {
      const int j = j0 + lane;
      len += __popcll(__ballot(j < staged && topk_keys[c * kk + j] >= thr_key));
    }
    if (len == staged && nc > staged) {
      for (int j0 = staged; j0 < nc; j0 += 64) {
        const int j = j0 + lane;
        const bool take = j < nc && f2key(keep[((size_t)c * m_stride + j) * 5 + 4]) >= thr_key;
        const unsigned long long tk = __ballot(take);
        const unsigned long long valid = nc - j0 >= 64 ? ~0ull : ((1ull << (nc - j0)) - 1ull);
        if (tk == valid) { len += __popcll(tk); continue; }
        len += __builtin_ctzll(~tk);  // non-increasing scores: the survivors end at the first row below the threshold
        break;
      }
    }
    if (lane == 0) cls_off[c] = len;
  }
  __syncthreads();
  if (wid == 0) {  // exclusive prefix sum of the survivor counts, 64 classes per step
    int run = 0;
    for (int c0 = 0; c0 < n_cls; c0 += 64) {
      const int c = c0 + lane;
      const int v = c < n_cls ? cls_off[c] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
      }
      const int tot = __shfl(incl, 63);
      if (c < n_cls) cls_off[c] = run + incl - v;
      run += tot;
    }
    if (lane == 0) cls_off[n_cls] = run;
  }
  __syncthreads();
  const int all = cls_off[n_cls];
  for (int o = tid; o < min(all, max_out); o += blockDim.x) {
    const int c = topk_class_of(cls_off, n_cls, o);
    const float *row = keep + ((size_t)c * m_stride + (o - cls_off[c])) * 5;
    float *q = out + 6 * (size_t)o;
    q[0] = row[0]; q[1] = row[1]; q[2] = row[2]; q[3] = row[3]; q[4] = row[4]; q[5] = (float)(c + 1);
  }
  if (tid == 0) *n_out = all;  // the UNTRUNCATED survivor count
}

// image.scale(src, W2, H2) 'bilinear' (external `image` rock, ImageDetect.lua:41; parity unpinned — see mpn.h):
// one output sample of a 1-D resample; `sstride` walks the source line.
__device__ __forceinline__ float scale_sample(const float *__restrict__ src, long sstride, long slen, long dlen, long d) {
  if (dlen > slen) {
    if (slen == 1) return src[0];
    if (d == dlen - 1) return src[(slen - 1) * sstride];
    const float scale = (float)(slen - 1) / (float)(dlen - 1);
    float sf = (float)d * scale;
    const long si = (long)sf;
    sf -= (float)si;
    return (1.0f - sf) * src[si * sstride] + sf * src[(si + 1) * sstride];
  } else if (dlen < slen) {
    const float scale = (float)slen / (float)dlen;
    const float s0 = (float)d * scale, s1 = (float)(d + 1) * scale;
    const long i0 = (long)s0, i1 = (long)s1;
    const float f0 = s0 - (float)i0, f1 = s1 - (float)i1;
    float acc = (1.0f - f0) * src[i0 * sstride], n = 1.0f - f0;
    for (long i = i0 + 1; i < i1; ++i) { acc += src[i * sstride]; n += 1.0f; }
    if (i1 < slen && i1 > i0) { acc += f1 * src[i1 * sstride]; n += f1; }
    return acc / n;
  }
  return src[d * sstride];
}
// pass 0: rows  [C,H,W] -> tmp [C,H,W2];  pass 1: columns tmp -> out [C,H2,W2]
__global__ void image_scale_kernel(const float *__restrict__ in, int C, int H, int W, int H2, int W2, int pass,
                                   float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pass == 0) {
    size_t total = (size_t)C * H * W2;
    if (t >= total) return;
    long x = (long)(t % W2); size_t r = t / W2;  // r = c*H + y
    out[t] = scale_sample(in + r * W, 1, W, W2, x);
  } else {
    size_t total = (size_t)C * H2 * W2;
    if (t >= total) return;
    long x = (long)(t % W2); size_t r = t / W2;
    long y = (long)(r % H2); size_t c = r / H2;
    out[t] = scale_sample(in + c * (size_t)H * W2 + x, W2, H, H2, y);
  }
}

// testCoco/init.lua:65-85 — detection wire rows for COCO evaluation: {image_id, x1-1, y1-1, x2-x1, y2-y1, score, category_id}
// from {x1,y1,x2,y2,score,class(1-based)} rows (mpn_keep_top_k's output); cat_ids maps class -> dataset category id.
__global__ void dets_to_coco_kernel(const float *__restrict__ dets, const int *__restrict__ n_dets, int max_n, float image_id,
                                    const float *__restrict__ cat_ids, int n_cat, float *__restrict__ rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = n_dets ? min(*n_dets, max_n) : max_n;
  if (i >= n) return;
  const float *d = dets + 6 * (size_t)i;
  float *o = rows + 7 * (size_t)i;
  int cls = (int)d[5];
  o[0] = image_id;
  o[1] = d[0] - 1.0f;
  o[2] = d[1] - 1.0f;
  o[3] = d[2] - d[0];
  o[4] = d[3] - d[1];
  o[5] = d[4];
  o[6] = (cat_ids && cls >= 1 && cls <= n_cat) ? cat_ids[cls - 1] : (float)cls;
}

// DataSetJSON.lua:157-239 — proposal table rows are {y1,x1,y2,x2}; loadROIDB filters by area (w*h > min_area, no +1) and
// emits {x1,y1,x2,y2} (the {2,1,4,3} column permute).  keep[i] = 1 when the row survives the area filter.
__global__ void proposals_permute_filter_kernel(const float *__restrict__ in, int n, float min_area, float *__restrict__ out,
                                                int *__restrict__ keep) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = in[4 * i], b = in[4 * i + 1], c = in[4 * i + 2], d = in[4 * i + 3];
  const float w0 = c - a, w1 = d - b;
  const float s = w0 * w1;
  out[4 * i] = b; out[4 * i + 1] = a; out[4 * i + 2] = d; out[4 * i + 3] = c;
  if (keep) keep[i] = (min_area == 0.0f || s > min_area) ? 1 : 0;
}

}  // namespace mpn

using namespace mpn;

extern "C" int mpn_image_scale(const float *d_in, int C, int H, int W, int H2, int W2, float *d_tmp, float *d_out, void *stream) {
  MPN_CHECK_ARG(d_in && d_tmp && d_out && C > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0);
  size_t t0 = (size_t)C * H * W2, t1 = (size_t)C * H2 * W2;
  hipLaunchKernelGGL(image_scale_kernel, dim3((unsigned)cdiv_sz(t0, 256)), dim3(256), 0, as_stream(stream), d_in, C, H, W, H2, W2, 0, d_tmp);
  MPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(image_scale_kernel, dim3((unsigned)cdiv_sz(t1, 256)), dim3(256), 0, as_stream(stream), d_tmp, C, H, W, H2, W2, 1, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_dets_to_coco_rows(const float *d_dets, const int *d_n_dets, int max_n, float image_id, const float *d_cat_ids,
                                     int n_cat, float *d_rows, void *stream) {
  MPN_CHECK_ARG(max_n >= 0);
  if (max_n == 0) return MPN_OK;
  MPN_CHECK_ARG(d_dets && d_rows);
  hipLaunchKernelGGL(dets_to_coco_kernel, dim3(cdiv(max_n, 256)), dim3(256), 0, as_stream(stream), d_dets, d_n_dets, max_n, image_id,
                     d_cat_ids, n_cat, d_rows);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_proposals_permute_filter(const float *d_in, int n, float min_area, float *d_out, int *d_keep, void *stream) {
  MPN_CHECK_ARG(n >= 0);
  if (n == 0) return MPN_OK;
  MPN_CHECK_ARG(d_in && d_out);
  hipLaunchKernelGGL(proposals_permute_filter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), d_in, n, min_area, d_out, d_keep);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_image_transform(const float *d_in, int H, int W, const int *h_swap, double scale,
                                   const double *h_mean, const double *h_std, float *d_out, void *stream) {
  MPN_CHECK_ARG(d_in && d_out && h_swap && h_mean && H > 0 && W > 0);
  for (int c = 0; c < 3; ++c) MPN_CHECK_ARG(h_swap[c] >= 0 && h_swap[c] < 3);
  MPN_CHECK_ARG(d_in != d_out);
  size_t plane = (size_t)H * W;
  dim3 grid((unsigned)cdiv_sz(plane, 256), 3);
  hipLaunchKernelGGL(image_transform_kernel, grid, dim3(256), 0, as_stream(stream), d_in, plane, h_swap[0], h_swap[1],
                     h_swap[2], scale, h_mean[0], h_mean[1], h_mean[2], h_std ? h_std[0] : 1.0, h_std ? h_std[1] : 1.0,
                     h_std ? h_std[2] : 1.0, h_std ? 1 : 0, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" double mpn_pick_scale(int H, int W, double target, double max_size) {
  double mn = H < W ? H : W, mx = H < W ? W : H;
  double s = target / mn;
  if (round(s * mx) > max_size) s = max_size / mx;
  return s;
}

extern "C" int mpn_project_im_rois(const float *d_boxes, int n, double scale, float *d_rois, void *stream) {
  MPN_CHECK_ARG(n >= 0);
  if (n == 0) return MPN_OK;
  MPN_CHECK_ARG(d_boxes && d_rois);
  hipLaunchKernelGGL(project_rois_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), d_boxes, n, (float)scale,
                     d_rois, static_cast<float *>(nullptr));
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

namespace mpn {
// mpn_project_im_rois + a verbatim copy of the boxes [n,4] (pipeline.hip: the deferred heads decode from it on the side stream)
int project_im_rois_copy(const float *d_boxes, int n, double scale, float *d_rois, float *d_boxes_copy, hipStream_t s) {
  MPN_CHECK_ARG(n >= 0 && scale > 0.0);
  if (n == 0) return MPN_OK;
  MPN_CHECK_ARG(d_boxes && d_rois && d_boxes_copy);
  hipLaunchKernelGGL(project_rois_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, d_boxes, n, (float)scale, d_rois, d_boxes_copy);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}
}  // namespace mpn

extern "C" int mpn_foveal_forward(const float *d_rois, int N, float *d_out, void *stream) {
  MPN_CHECK_ARG(N >= 0);
  if (N == 0) return MPN_OK;
  MPN_CHECK_ARG(d_rois && d_out);
  hipLaunchKernelGGL(foveal_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), d_rois, N, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_context_region_forward(const float *d_rois, int N, double scale, float *d_out, void *stream) {
  MPN_CHECK_ARG(N >= 0);
  if (N == 0) return MPN_OK;
  MPN_CHECK_ARG(d_rois && d_out);
  float a = (float)((1.0 + scale) / 2.0), b = (float)((1.0 - scale) / 2.0);
  hipLaunchKernelGGL(context_region_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), d_rois, N, a, b, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_bbox_norm_forward(float *d_bbox, int N, int C4, const float *h_mean4, const float *h_std4,
                                     void *stream) {
  MPN_CHECK_ARG(N >= 0 && C4 >= 0 && (C4 % 4) == 0 && h_mean4 && h_std4);
  size_t total = (size_t)N * C4;
  if (total == 0) return MPN_OK;
  MPN_CHECK_ARG(d_bbox);
  hipLaunchKernelGGL(bbox_norm_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, as_stream(stream), d_bbox, total,
                     h_mean4[0], h_mean4[1], h_mean4[2], h_mean4[3], h_std4[0], h_std4[1], h_std4[2], h_std4[3]);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_select_boxes_forward(const float *d_scores, const float *d_bbox, int N, int C, float *d_out,
                                        void *stream) {
  MPN_CHECK_ARG(N >= 0 && C > 0);
  if (N == 0) return MPN_OK;
  MPN_CHECK_ARG(d_scores && d_bbox && d_out);
  hipLaunchKernelGGL(select_boxes_kernel, dim3(cdiv(N, 256)), dim3(256), 0, as_stream(stream), d_scores, d_bbox, N, C,
                     d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_softmax_forward(const float *d_x, int M, int C, float *d_y, void *stream) {
  MPN_CHECK_ARG(M >= 0 && C > 0);
  if (M == 0) return MPN_OK;
  MPN_CHECK_ARG(d_x && d_y);
  hipLaunchKernelGGL(softmax_kernel, dim3(cdiv(M, 4)), dim3(256), 0, as_stream(stream), d_x, M, C, d_y);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

namespace mpn {
int launch_bbox_decode(const float *d_boxes, const float *d_deltas, int N, int C, float *d_out, int clamp, float im_w,
                       float im_h, hipStream_t s) {
  size_t total = (size_t)N * C;
  hipLaunchKernelGGL(bbox_decode_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_boxes, d_deltas, N, C,
                     d_out, clamp, im_w, im_h);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}
}  // namespace mpn

extern "C" int mpn_bbox_decode(const float *d_boxes, const float *d_deltas, int N, int C, float *d_out, void *stream) {
  MPN_CHECK_ARG(N >= 0 && C > 0);
  if (N == 0) return MPN_OK;
  MPN_CHECK_ARG(d_boxes && d_deltas && d_out);
  return launch_bbox_decode(d_boxes, d_deltas, N, C, d_out, 0, 0.f, 0.f, as_stream(stream));
}

extern "C" int mpn_clamp_boxes(float *d_bbox, size_t n_pairs, float im_w, float im_h, void *stream) {
  if (n_pairs == 0) return MPN_OK;
  MPN_CHECK_ARG(d_bbox);
  hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)cdiv_sz(n_pairs, 256)), dim3(256), 0, as_stream(stream), d_bbox, n_pairs,
                     im_w, im_h);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_select_scored(const float *d_scores, const float *d_bbox, int N, int C, int first_cls, float thresh,
                                 float *d_scored, int *d_counts, int *d_src_idx, void *stream) {
  MPN_CHECK_ARG(N >= 0 && C > 0 && first_cls >= 0 && first_cls <= C && d_counts);
  int n_cls = C - first_cls;
  if (n_cls == 0) return MPN_OK;
  if (N == 0) {
    MPN_CHECK_HIP(hipMemsetAsync(d_counts, 0, sizeof(int) * n_cls, as_stream(stream)));
    return MPN_OK;
  }
  MPN_CHECK_ARG(d_scores && d_bbox && d_scored);
  hipLaunchKernelGGL(select_scored_kernel, dim3(n_cls), dim3(256), 0, as_stream(stream), d_scores, d_bbox, N, C, first_cls,
                     thresh, d_scored, d_counts, d_src_idx);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_keep_top_k(const float *d_keep, const int *d_n_keep, int n_cls, int m_stride, int k, float *d_thresh,
                              float *d_out, int max_out, int *d_n_out, void *stream) {
  MPN_CHECK_ARG(n_cls >= 0 && m_stride >= 0 && k > 0 && max_out >= 0);
  MPN_CHECK_ARG(d_thresh && d_n_out && (max_out == 0 || d_out));
  MPN_CHECK_ARG(n_cls == 0 || (d_keep && d_n_keep));
  MPN_CHECK_ARG(n_cls <= kTopkMaxCls);
  size_t nkeys = (size_t)n_cls * (size_t)m_stride;
  if (nkeys > (size_t)kTopkLdsKeys) nkeys = kTopkLdsKeys;
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(keep_top_k_kernel), kTopkLdsKeys * 4 + 32 + kTopkCand * 4); if (rc_attr) return rc_attr; }
  hipLaunchKernelGGL(keep_top_k_kernel, dim3(1), dim3(1024), nkeys * 4 + 32 + kTopkCand * 4, as_stream(stream), d_keep, d_n_keep, n_cls, m_stride, k,
                     d_thresh, d_out, max_out, d_n_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_keep_top_k_sorted(const float *d_keep, const int *d_n_keep, int n_cls, int m_stride, int k, float *d_thresh,
                                     float *d_out, int max_out, int *d_n_out, void *stream) {
  MPN_CHECK_ARG(n_cls >= 0 && m_stride >= 0 && k > 0 && max_out >= 0);
  MPN_CHECK_ARG(d_thresh && d_n_out && (max_out == 0 || d_out));
  MPN_CHECK_ARG(n_cls == 0 || (d_keep && d_n_keep));
  MPN_CHECK_ARG(n_cls <= kTopkMaxCls);
  const size_t slots = (size_t)n_cls * (size_t)(k < m_stride ? k : m_stride);
  if (slots > (size_t)kTopkLdsKeys || slots == 0)  // more candidates than the LDS stage holds: the general kernel is as good
    return mpn_keep_top_k(d_keep, d_n_keep, n_cls, m_stride, k, d_thresh, d_out, max_out, d_n_out, stream);
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(keep_top_k_sorted_kernel), kTopkLdsKeys * 4); if (rc_attr) return rc_attr; }
  hipLaunchKernelGGL(keep_top_k_sorted_kernel, dim3(1), dim3(1024), slots * 4, as_stream(stream), d_keep, d_n_keep, n_cls, m_stride, k, d_thresh, d_out,
                     max_out, d_n_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

#ifdef MPN_DEBUG_HOOKS
extern "C" int mpn_debug_get_topk_trace(unsigned long long *h_out8) {
  MPN_CHECK_ARG(h_out8);
  MPN_CHECK_HIP(hipMemcpyFromSymbol(h_out8, HIP_SYMBOL(mpn::g_topk_trace), 8 * sizeof(unsigned long long)));
  return MPN_OK;
}
#endif
