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
                                                      int m_cap, const int *__restrict__ flags, float *__restrict__ gwork) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (flags && flags[blockIdx.x] != 2) return;  // this class took a sorted/bitmask path
  // the class's six working arrays: LDS, or (tables wider than MPN_NMS_MAX_BOXES — nms.c itself has no size limit) a
  // per-class slice of HBM scratch; one wave owns them, __syncthreads() orders its lanes' accesses at workgroup scope
  float *base = gwork ? gwork + (size_t)blockIdx.x * 6 * m_cap : lds;
  float *X1 = base, *Y1 = base + m_cap, *X2 = base + 2 * m_cap, *Y2 = base + 3 * m_cap, *S = base + 4 * m_cap;
  int *POS = reinterpret_cast<int *>(base + 5 * m_cap);

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


// =================================================================================================
// Fast path: classes whose scores are all distinct (the common case).  With no bit-equal scores the
// reference's pick order is simply "descending score", so the greedy loop factors into three
// data-parallel / latency-short phases (the exact wave kernel above stays the fallback for classes
// with ties, NaNs, or more boxes than the sort's LDS budget):
//   1. sort_kernel   one block per class: LDS bitonic sort of 64-bit keys (score desc, index asc);
//                    writes boxes in rank order, detects ties -> flags[c] = 1 (fallback).
//   2. mask_kernel   whole GPU: 64-bit suppression words  mask[c][i][w] bit j = IoU(rank i, rank 64w+j) > thr
//                    for j > i (upper triangle only), IoU evaluated exactly as nms.c:14-41.
//   3. scan_kernel   one wavefront per class walks ranks in 64-box chunks: the chunk's diagonal word is
//                    resolved with scalar bit tricks (ctz / readlane), then the rows of the kept boxes —
//                    speculatively loaded for the whole chunk, 64 independent loads in flight — are OR-ed
//                    into the per-lane `removed` words.
// =================================================================================================
__device__ __forceinline__ unsigned nms_f2key(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int kSortMax = 8192;  // 64 KiB of LDS keys
constexpr int kTieMax = 4096;   // nms_tie_kernel: one 64-bit alive/occupancy word per lane
constexpr int kTieLdsMask = 1024;  // full symmetric mask kept in LDS up to this many boxes (128 KiB)
// Classes with up to this many tied adjacent pairs take the chunked scan with the lazy, vectorised position replay (flag 3): its
// cost is the tie-free scan plus one batched update of the slot model per equal-score run that still has two alive members at
// its turn (every update re-loads register windows: the cost grows with the number of such runs).  (Round 2's eager form — head replay and death recording inside every chunk — measured 850 k cycles against the slot
// kernel's 320 k and was not dispatched to.)  Classes with more ties keep the slot-emulating kernel, whose per-pick cost does
// not depend on the number of ties.
constexpr int kFewTies = 12;  // measured (tools/nms_trace.py, 1000 boxes): 4 tied pairs 189 k cycles, 16 pairs 680 k = the slot kernel's cost

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float *__restrict__ scored, const int *__restrict__ counts,
                                                        int m_stride, float4 *__restrict__ sbox, float *__restrict__ sscore,
                                                        int *__restrict__ sidx, int *__restrict__ n_sel,
                                                        int *__restrict__ flags, int force_mode) {
  // flags[c]: 0 = tie-free -> chunked scan; 1 = ties -> nms_tie_kernel (exact slot emulation on the
  // bitmask); 2 = NaN scores or too many boxes -> nms_wave_kernel (exact IoU sweep)
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ int bad, nsel, hasnan;  // bad = number of bit-equal adjacent score pairs
  const int cls = blockIdx.x, tid = threadIdx.x;
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  int n_pad = 64;
  while (n_pad < m) n_pad <<= 1;
  if (tid == 0) { bad = 0; nsel = 0; hasnan = 0; }
  __syncthreads();
  if (m > kSortMax || force_mode == 1) { if (tid == 0) { flags[cls] = 2; n_sel[cls] = 0; } return; }
  const bool dense = force_mode == 4;  // utils.nms_dense: picks follow the SORT's order (no position history), every box is pickable
  const float *src = scored + (size_t)cls * m_stride * 5;
  for (int i = tid; i < n_pad; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < m) {
      float s = src[5 * (size_t)i + 4];
      if (s != s) hasnan = 1;  // NaN: let the exact sweep kernel reproduce the reference's behaviour
      k = ((unsigned long long)(~nms_f2key(s)) << 32) | (unsigned)i;
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= n_pad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < n_pad / 2; t += blockDim.x) {
        int lo = (t / j) * 2 * j + (t % j), hi = lo + j;
        bool up = ((lo & k) == 0);
        unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  float4 *ob = sbox + (size_t)cls * m_stride;
  float *os = sscore + (size_t)cls * m_stride;
  int *oi = sidx + (size_t)cls * m_stride;
  int local_sel = 0;
  for (int p = tid; p < m; p += blockDim.x) {
    unsigned long long k = keys[p];
    int i = (int)(unsigned)k;
    if (p + 1 < m && (unsigned)(keys[p + 1] >> 32) == (unsigned)(k >> 32)) atomicAdd(&bad, 1);  // bit-equal scores
    const float *r = src + 5 * (size_t)i;
    float s = r[4];
    ob[p] = make_float4(r[0], r[1], r[2], r[3]);
    os[p] = s;
    oi[p] = i;
    if (dense || s > -10000000.0f) ++local_sel;  // nms.c:75: never picked otherwise; sorted => a prefix
  }
  atomicAdd(&nsel, local_sel);
  __syncthreads();
  if (tid == 0) {
    // 0 = tie-free, 3 = a few tied pairs (the chunked scan with position replay), 1 = many ties (slot emulation on the
    // LDS-staged bitmask), 2 = NaN scores (exact sweep)
    int f = hasnan ? 2 : (force_mode == 2 ? 1 : (force_mode == 3 ? 3 : (bad == 0 ? 0 : (bad <= kFewTies ? 3 : 1))));
    // the host launches nms_tie_kernel only when m_stride <= kTieMax (its LDS tables are sized by m_stride): a class with
    // ties in a wider table goes to the exact sweep kernel, whatever its own count
    if (f == 1 && m_stride > kTieMax) f = 2;
    if (dense) f = 0;
    flags[cls] = f;
    n_sel[cls] = nsel;
  }
}

// utils.nms_dense's suppression test (utils.lua:430-449), fp32 operation for operation: `c` = the picked box, `o` = the other.
//   xx1:copy(x1):clamp(x1[c], huge) ... xx2:copy(x2):clamp(0, x2[c]); w = clamp(xx2 + (-1)*xx1 + 1, 0, huge); inter = w*h;
//   union = area + (-1)*inter + area[c]; ol = inter / union; suppressed where ol > overlap.   THTensor_(clamp) = v<lo ? lo : (v>hi ? hi : v)
__device__ __forceinline__ bool dense_suppresses(const float4 c, const float4 o, float overlap) {
  const float xx1 = o.x < c.x ? c.x : o.x, yy1 = o.y < c.y ? c.y : o.y;
  const float xx2 = o.z < 0.0f ? 0.0f : (o.z > c.z ? c.z : o.z), yy2 = o.w < 0.0f ? 0.0f : (o.w > c.w ? c.w : o.w);
  float w = xx2 - xx1; w = w + 1.0f; w = w < 0.0f ? 0.0f : w;
  float h = yy2 - yy1; h = h + 1.0f; h = h < 0.0f ? 0.0f : h;
  const float inter = w * h;
  const float ao = ((o.z - o.x) + 1.0f) * ((o.w - o.y) + 1.0f), ac = ((c.z - c.x) + 1.0f) * ((c.w - c.y) + 1.0f);
  float uni = ao - inter; uni = uni + ac;
  return inter / uni > overlap;
}

template <bool DENSE>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float4 *__restrict__ sbox, const int *__restrict__ n_sel,
                                                       const int *__restrict__ flags, const int *__restrict__ counts,
                                                       int m_stride, int w64, float thr,
                                                       unsigned long long *__restrict__ mask) {
  __shared__ float4 cols[64];
  const int cls = blockIdx.z;
  const int flag = flags[cls];
  if (flag == 2) return;
  const bool full = flag == 1 || flag == 3;  // tie classes: full symmetric rows over ALL boxes (unpickable ones can still be suppressed)
  const int n = n_sel[cls];     // rows: only pickable ranks ever suppress
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  const int ncol = full ? m : n;
  const int nrow = full ? m : n;  // full: the lazy replay (nms_scan_kernel) reads the row of ANY box, pickable or not, to date its death
  const int w = blockIdx.x, rb = blockIdx.y;
  if (w * 64 >= ncol || rb * 256 >= nrow) return;
  if (!full && rb * 256 > w * 64 + 63) return;  // strictly lower triangle: never read by the chunked scan
  const float4 *b = sbox + (size_t)cls * m_stride;
  if (threadIdx.x < 64) {
    int j = w * 64 + threadIdx.x;
    cols[threadIdx.x] = j < ncol ? b[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int i = rb * 256 + threadIdx.x;
  if (i >= nrow) return;
  const float4 a = b[i];
  unsigned long long bits = 0;
  const int jn = min(64, ncol - w * 64);
  for (int jj = 0; jj < jn; ++jj) {
    const int j = w * 64 + jj;
    // the chunk's own (diagonal) word is always symmetric: the scan resolves a chunk by a fixpoint over "no kept lower rank
    // overlaps me", which reads the lower triangle of that word
    if ((full || (i >> 6) == w) ? (j == i) : (j <= i)) continue;
    const float4 c = cols[jj];
    if constexpr (DENSE) {  // the picked box is the LOWER rank of the pair (the diagonal word's lower triangle is read as "a kept lower rank overlaps me")
      if (j > i ? dense_suppresses(a, c, thr) : dense_suppresses(c, a, thr)) bits |= 1ull << jj;
    } else {
      float iou = iou_plus1(a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w);  // overlap(best, other), nms.c:92
      if (!(iou <= thr)) bits |= 1ull << jj;
    }
  }
  mask[((size_t)cls * m_stride + i) * w64 + w] = bits;
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int l) {
  unsigned lo = __shfl((unsigned)v, l), hi = __shfl((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
  unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// Chunked scan, with the reference's position history replayed — lazily — when the class has (a few) bit-equal scores.
//
// Tie-free class (flag 0): the pick order is the rank order, a chunk of 64 ranks is resolved with wave-uniform bit
// arithmetic on its diagonal word and the kept rows are OR-ed into the per-lane `removed` words.
//
// Class with a few tied pairs (flag 3): nms.c picks, among bit-equal scores, the box that sits FIRST in its array
// (nms.c:74-81), and the array is permuted by every round — the old first element takes the picked box's place
// (nms.c:83-85), the survivors keep their order (nms.c:91-98).  Which tied box comes first therefore depends on the whole
// history.  Positions ("slots") only ever matter when an equal-score run has TWO OR MORE alive members at its turn, so:
//   * everything else is resolved exactly as in the tie-free case, and all that is recorded per pick is its round
//     (rnd[rank], klist[round]) — no per-pick position work on the scan's critical path;
//   * when such a run comes up, the slot model is first brought up to date for all rounds since the last update
//     (`simulate`): a round's head is the first slot whose occupant is alive at that round; a round vacates the head's slot
//     and moves boxes only to LATER slots, so one pointer sweeps the slots once per class, 64 slots per register window.
//     Within a window the heads of consecutive rounds are a 64-lane fixpoint
//         taken_l = valid_l && death_l >= t0 + popcount(taken & lanes_below_l)
//     (a lane's round = the first pending round + the heads before it), solved with a handful of ballots; the moves of the
//     whole batch are then two LDS scatters.  The batch is cut short where a move lands inside the window ahead of the head,
//     or where a round's pick was itself moved earlier in the batch (its slot in LDS would be stale);
//   * "alive at round t" is a comparison with the occupant's death round, computed on demand for the 64 window occupants:
//     a picked box dies in its own round, a suppressed one in the round of the first pick that overlaps it = the lowest
//     kept rank in its (symmetric, flag 3) mask row — or the earliest-picked member of that rank's equal-score run;
//   * the run's pick is its alive member with the smallest slot (nms.c:77-80's strict '>' keeps the first maximum).
// The algorithm was validated against the reference's compiled nms.c in a Python model first (tools/models/
// nms_lazy_replay_model.py).  LDS (flag 3 only): int16 pos[rank] / occ[slot] / rnd[rank] / klist[round] / mv[rank].
constexpr int kForever = 0x3fffffff;
template <int WPL>  // 64-bit `removed` words per lane: covers m <= 4096 * WPL
__global__ __launch_bounds__(64) void nms_scan_kernel(const float4 *__restrict__ sbox, const float *__restrict__ sscore,
                                                      const int *__restrict__ sidx, const int *__restrict__ n_sel,
                                                      int *__restrict__ flags, const int *__restrict__ counts, int m_stride, int w64,
                                                      const unsigned long long *__restrict__ mask, float *__restrict__ keep,
                                                      int *__restrict__ keep_idx, int *__restrict__ n_keep, int m_cap,
                                                      unsigned long long *__restrict__ trace, int guard_limit) {
  // guard_limit: bound of the replay's two progress loops (0 = 2 m_cap + slack, which a correct run cannot reach).  Should a loop ever
  // run into its bound, the class is NOT emitted truncated: its flag becomes 2 and the exact IoU-sweep kernel, launched after this one,
  // redoes it (tests force the path with a tiny limit).
  bool failed = false;
  extern __shared__ __attribute__((aligned(16))) short lds16[];  // pos | occ | rnd | klist | mv, m_cap each   (flag 3)
  // tools/nms_trace.py (debug flavour): s_memtime stamps of class 0's chunks -> trace[c * 8 + k]
#define NMS_STAMP(k) do { if (MPN_ABLATE(trace != nullptr) && blockIdx.x == 0 && lane == 0 && c < 64) trace[c * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
  __shared__ unsigned long long tiew[64 * WPL];
  const int cls = blockIdx.x, lane = threadIdx.x;
  const int flag = flags[cls];
  if (flag != 0 && flag != 3) return;
  const bool replay = flag == 3;
  const int n = n_sel[cls];
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  const float4 *b = sbox + (size_t)cls * m_stride;
  const float *sc = sscore + (size_t)cls * m_stride;
  const int *si = sidx + (size_t)cls * m_stride;
  const unsigned long long *mk = mask + (size_t)cls * m_stride * w64;
  float *kout = keep + (size_t)cls * m_stride * 5;
  int *kidx = keep_idx ? keep_idx + (size_t)cls * m_stride : nullptr;
  short *pos = lds16, *occ = lds16 + m_cap, *rnd = lds16 + 2 * m_cap, *klist = lds16 + 3 * m_cap, *mv = lds16 + 4 * m_cap;
  unsigned long long removed[WPL], keptw[WPL];
#pragma unroll
  for (int h = 0; h < WPL; ++h) { removed[h] = 0; keptw[h] = 0; }
  if (replay) {
    for (int r = lane; r < m; r += kWave) {
      const int x = si[r];
      pos[r] = (short)x; occ[x] = (short)r; rnd[r] = 0; mv[r] = 0;
    }
    for (int r0 = 0; r0 < 64 * 64 * WPL; r0 += kWave) {  // bit r of the tie words: ranks r and r+1 are pickable and carry the same score
      const int r = r0 + lane;
      const bool tie = (r + 1 < n) && (sc[r] == sc[r + 1]);
      const unsigned long long bal = __ballot(tie);
      if (lane == 0) tiew[r0 >> 6] = bal;
      if (r0 + 64 >= m) {  // the remaining words are zero
        for (int w = (r0 >> 6) + 1 + lane; w < 64 * WPL; w += kWave) tiew[w] = 0ull;
        break;
      }
    }
    __syncthreads();
  }
  auto word_of = [&](int w) -> unsigned long long {  // removed word w (wave-uniform index)
    unsigned long long v = readlane64(removed[0], w & 63);
    if constexpr (WPL > 1) { if (w >= 64) v = readlane64(removed[1], w & 63); }
    return v;
  };
  auto kept_word = [&](int w) -> unsigned long long {  // kept word w (wave-uniform index)
    unsigned long long v = readlane64(keptw[0], w & 63);
    if constexpr (WPL > 1) { if (w >= 64) v = readlane64(keptw[1], w & 63); }
    return v;
  };
  auto removed_bit = [&](int r) -> bool {  // per-lane rank; every lane of the wave must call it
    const int w = r >> 6;
    unsigned long long rw = shfl64(removed[0], w & 63);
    if constexpr (WPL > 1) { const unsigned long long r1 = shfl64(removed[1], w & 63); if (w >= 64) rw = r1; }
    return (rw >> (r & 63)) & 1ull;
  };
  // ---- the slot model, brought up to date on demand (flag 3) ----
  int sim_done = 0, hp = 0, bid = 0;  // rounds replayed so far; slots < hp are vacated or hold dead boxes; batch stamp
  int wf = -1, wd = -1;               // register window: occupant (rank) and its death round of slot W0 + lane
  const int nw_kept = (n + 63) >> 6;  // picks are ranks < n
  // death round of rank f (per lane; f < 0 -> -1): its own round if it was picked; kForever while it is alive; else the round of
  // the first pick that overlaps it.  Every lane must call it (cross-lane reads inside).
  auto death_of = [&](int f) -> int {
    const int fs = f < 0 ? 0 : f;
    const int rk = (int)rnd[fs];
    const bool rem = removed_bit(fs);
    int d = f < 0 ? -1 : (rk > 0 ? rk : (rem ? 0 : kForever));
    bool need = d == 0;
    for (int w0 = 0; w0 < nw_kept; w0 += 16) {
      if (!__ballot(need)) break;
      unsigned long long x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = (need && w0 + q < nw_kept) ? mk[(size_t)fs * w64 + w0 + q] : 0ull;  // independent loads, all in flight: one round trip per 1024 ranks
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int w = w0 + q;
        if (w < nw_kept) {  // wave-uniform
          const unsigned long long hit = x[q] & kept_word(w);
          if (need && hit) {
            int bb = w * 64 + __builtin_ctzll(hit);
            int best = (int)rnd[bb];
            while ((tiew[bb >> 6] >> (bb & 63)) & 1ull) {  // the rest of that rank's equal-score run (its picks may be out of rank order)
              ++bb;
              const unsigned long long rwd = (bb >> 6) == w ? x[q] : mk[(size_t)fs * w64 + (bb >> 6)];
              const int rb = (int)rnd[bb];
              if (((rwd >> (bb & 63)) & 1ull) && rb > 0 && rb < best) best = rb;
            }
            d = best;
            need = false;
          }
        }
      }
    }
    return d;
  };
  auto simulate = [&](int t1) {  // replay rounds sim_done + 1 .. t1 (all picked: klist / rnd hold them)
    int loaded = -64;  // the window in registers (death rounds are fixed for the duration of a call: every pending round is decided)
    const int glim = guard_limit > 0 ? guard_limit : 2 * m_cap + 1024;
    for (int guard = 0; sim_done < t1; ++guard) {
      if (guard >= glim) { failed = true; break; }
      __syncthreads();  // LDS writes of the previous batch / of the picks since the last call
      const int W0 = hp & ~63, p = hp - W0;
      if (W0 != loaded) { const int sl = W0 + lane; wf = sl < m ? (int)occ[sl] : -1; wd = death_of(wf); loaded = W0; }
      const int t0 = sim_done + 1;
      const unsigned long long below = (1ull << lane) - 1ull;
      const bool cand = lane >= p && wf >= 0;
      unsigned long long taken = 0ull;
      for (int it = 0; it < 66; ++it) {  // lane l's answer depends on lower lanes only: settles in <= 64 rounds, typically a few
        const int rd = t0 + __popcll(taken & below);
        const unsigned long long nt = __ballot(cand && wd >= rd && rd <= t1);
        if (nt == taken) break;
        taken = nt;
      }
      if (!taken) { hp = W0 + 64; if (hp >= m) break; continue; }  // window exhausted
      const bool mine = (taken >> lane) & 1ull;
      const int t = t0 + __popcll(taken & below);
      const int pick = mine ? (int)klist[t - 1] : 0;
      const bool move = mine && wf != pick;  // boxes[0] <-> boxes[best] (nms.c:83-85); a no-op when the head IS the pick
      const int sb = mine ? (int)pos[pick] : 0;
      ++bid;
      if (move) mv[wf] = (short)bid;
      __syncthreads();
      const bool h1 = mine && mv[pick] == (short)bid;  // this round's pick was moved earlier in the batch: `sb` is stale
      const unsigned long long h1m = __ballot(h1), h2m = __ballot(move && sb < W0 + 64);  // h2: the move lands inside the window
      int cut = 64;  // commit the taken lanes below `cut`
      if (h1m) cut = __builtin_ctzll(h1m);
      if (h2m) { const int c2 = __builtin_ctzll(h2m) + 1; if (c2 < cut) cut = c2; }
      unsigned long long cm = cut >= 64 ? taken : (taken & ((1ull << cut) - 1ull));
      if (!cm) cm = taken & (~taken + 1ull);  // (cannot happen: the lowest taken lane has no earlier mover) — never stall
      if (((cm >> lane) & 1ull) && move) { occ[sb] = (short)wf; pos[wf] = (short)sb; }
      if (h2m & cm) {  // the committed move that landed inside the window: its slot now holds the moved box (same window, no reload)
        const int lc = __builtin_ctzll(h2m & cm);
        const int f2 = __builtin_amdgcn_readlane(wf, lc), d2 = __builtin_amdgcn_readlane(wd, lc), s2 = __builtin_amdgcn_readlane(sb, lc);
        if (lane == s2 - W0) { wf = f2; wd = d2; }
      }
      sim_done += __popcll(cm);
      hp = W0 + (64 - __builtin_clzll(cm));  // one past the highest committed lane
    }
    __syncthreads();
  };
  int kept = 0;
  const int nchunks = (n + 63) >> 6;
  unsigned long long tr_sim = 0ull, tr_t0 = MPN_ABLATE(trace != nullptr) ? __builtin_amdgcn_s_memtime() : 0ull;  // tools/nms_trace.py
  int tr_calls = 0;
  // the chunk's diagonal word, one row per lane, fetched a chunk ahead (its latency would otherwise sit in front of every chunk)
  unsigned long long diag_cur = (lane < n) ? mk[(size_t)lane * w64] : 0ull;
  for (int c = 0; c < nchunks; ++c) {
    if (failed) break;
    const int base = c << 6;
    const unsigned long long diag = diag_cur;
    if (c + 1 < nchunks) diag_cur = (base + 64 + lane < n) ? mk[(size_t)(base + 64 + lane) * w64 + (c + 1)] : 0ull;
    const int nv = min(64, n - base);
    const unsigned long long valid = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
    // A chunk is resolved in passes: alive ranks below the first alive rank that carries a tie bit go through the tie-free
    // rule (bit arithmetic on the diagonal word, rows folded in a batch); a first alive rank WITH a tie bit is one pick by
    // the exact rule; repeat until the chunk is done.
    const int plim = guard_limit > 0 ? guard_limit : 2 * m_cap + 130;
    for (int pass = 0;; ++pass) {  // (every pass picks a box or finishes the chunk; a run's picks may lie in LATER chunks)
      if (pass >= plim || failed) { failed = true; break; }
      const unsigned long long rem_c = word_of(c);
      unsigned long long alive_all = ~rem_c & valid;
      if (!alive_all) break;
      unsigned long long tcw = 0ull;
      if (replay) tcw = tiew[c];
      const unsigned long long tiesel = alive_all & tcw;
      const int first = __builtin_ctzll(alive_all);
      if (tiesel && __builtin_ctzll(tiesel) == first) {
        // ---- one pick by the exact rule: among the alive members of r0's equal-score run, the one sitting first in the array
        const int r0 = base + first;
        int e = r0;  // last rank of the run = the first rank >= r0 whose tie bit is clear
        for (;;) {
          const unsigned long long ones = tiew[e >> 6] >> (e & 63);
          const int span = 64 - (e & 63);
          const int cnt = (~ones) ? __builtin_ctzll(~ones) : 64;
          if (cnt < span) { e += cnt; break; }
          e += span;
          if (e >= n) { e = n - 1; break; }
        }
        // alive members: how many, and the lowest rank
        int n_alive = 0, low = 0x7fffffff;
        for (int rr = r0; rr <= e; rr += kWave) {
          const int r = rr + lane;
          const bool in = r <= e;
          const bool ok = !removed_bit(in ? r : r0) && in;
          n_alive += __popcll(__ballot(ok));
          if (ok && r < low) low = r;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(low, off); if (o < low) low = o; }
        int pick = __builtin_amdgcn_readfirstlane(low);
        if (n_alive >= 2) {  // only now do positions matter: bring the slot model up to date, then take the smallest slot
          const unsigned long long ts0 = MPN_ABLATE(trace != nullptr) ? __builtin_amdgcn_s_memtime() : 0ull;
          simulate(kept);
          if (MPN_ABLATE(trace != nullptr)) { tr_sim += __builtin_amdgcn_s_memtime() - ts0; ++tr_calls; }
          int bp = 0x7fffffff, br = -1;
          for (int rr = r0; rr <= e; rr += kWave) {
            const int r = rr + lane;
            const bool in = r <= e;
            const bool ok = !removed_bit(in ? r : r0) && in;
            const int pp = ok ? (int)pos[r] : 0x7fffffff;
            if (pp < bp) { bp = pp; br = r; }
          }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const int op = __shfl_xor(bp, off), orr = __shfl_xor(br, off);
            if (op < bp) { bp = op; br = orr; }
          }
          pick = __builtin_amdgcn_readfirstlane(br);
        }
        if (lane == 0) {
          const float4 bx = b[pick];
          float *q = kout + (size_t)kept * 5;
          q[0] = bx.x; q[1] = bx.y; q[2] = bx.z; q[3] = bx.w; q[4] = sc[pick];
          if (kidx) kidx[kept] = si[pick];
          klist[kept] = (short)pick;
          rnd[pick] = (short)(kept + 1);
        }
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
          const int w = lane + 64 * h;
          if (w < w64 && w * 64 < m) {
            removed[h] |= mk[(size_t)pick * w64 + w];
            if (w == (pick >> 6)) { removed[h] |= 1ull << (pick & 63); keptw[h] |= 1ull << (pick & 63); }
          }
        }
        ++kept;
        continue;
      }
      // ---- tie-free rule for the alive ranks below `limit`
      NMS_STAMP(0);
      const int limit = tiesel ? __builtin_ctzll(tiesel) : 64;
      const unsigned long long lim_mask = limit == 64 ? ~0ull : ((1ull << limit) - 1ull);
      // speculative: the rows of this range, this lane's word(s) — independent loads, all in flight
      unsigned long long rows[64];
      if (lane > c && lane < w64) {  // unconditional loads off one base pointer (rows past n / the range are read but never used)
        const unsigned long long *rp = mk + (size_t)base * w64 + lane;
#pragma unroll
        for (int r = 0; r < 64; ++r) rows[r] = rp[(size_t)r * w64];
      } else {
#pragma unroll
        for (int r = 0; r < 64; ++r) rows[r] = 0ull;
      }
      // resolve the range: rank l is kept iff it is alive and no KEPT lower rank overlaps it.  That recurrence has a unique
      // solution, so any fixpoint of "kept_l = alive_l && !(lower overlapping ranks & kept)" is the greedy answer; starting from
      // kept = alive it settles in (longest suppression chain) rounds of one ballot each, all 64 ranks at once — a serial
      // ctz / readlane walk measured ~150 cycles per kept box, a static 64-step SALU walk 85 per rank.
      unsigned long long keptmask, diag_acc;
      {
        const unsigned long long alive0 = alive_all & lim_mask;
        const unsigned long long lower = diag & ((1ull << lane) - 1ull);   // the symmetric diagonal word: lower ranks that overlap me
        const bool alive_l = (alive0 >> lane) & 1ull;
        NMS_STAMP(1);
        keptmask = alive0;
        for (;;) {
          const unsigned long long kn = __ballot(alive_l && !(lower & keptmask));
          if (kn == keptmask) break;
          keptmask = kn;
        }
        diag_acc = __ballot((diag & keptmask) != 0ull);  // every rank of the chunk overlapped by a kept one
      }
      // emit kept boxes in rank order
      NMS_STAMP(2);
      const bool mine = (keptmask >> lane) & 1ull;
      if (mine) {
        const int o = kept + __popcll(keptmask & ((1ull << lane) - 1ull));
        const float4 bx = b[base + lane];
        float *q = kout + (size_t)o * 5;
        q[0] = bx.x; q[1] = bx.y; q[2] = bx.z; q[3] = bx.w; q[4] = sc[base + lane];
        if (kidx) kidx[o] = si[base + lane];
        if (replay) { klist[o] = (short)(base + lane); rnd[base + lane] = (short)(o + 1); }  // all the slot model will need of these rounds
      }
      // fold the kept rows into `removed`: words after this chunk, and the chunk's own word (resolved range + in-chunk kills)
      NMS_STAMP(3);
      {
        unsigned long long acc = 0ull;
#pragma unroll
        for (int r = 0; r < 64; ++r) acc |= ((keptmask >> r) & 1ull) ? rows[r] : 0ull;
        removed[0] |= acc;
      }
      if (lane == c) { removed[0] |= (valid & lim_mask) | (diag_acc & valid); keptw[0] |= keptmask; }
      if constexpr (WPL > 1) {  // second word per lane (m > 4096): non-speculative, batched
        const int wsec = lane + 64;
        if (wsec < w64 && wsec > c) {
          unsigned long long km = keptmask;
          while (km) {
            const int r = __builtin_ctzll(km);
            km &= km - 1;
            removed[1] |= mk[(size_t)(base + r) * w64 + wsec];
          }
        }
        if (wsec == c) { removed[1] |= (valid & lim_mask) | (diag_acc & valid); keptw[1] |= keptmask; }
      }
      NMS_STAMP(4);
      NMS_STAMP(5);
      kept += __popcll(keptmask);
      if (limit == 64) break;
    }
  }
  if (failed) {  // a progress bound was hit: hand the class to the exact sweep (nms_wave_kernel checks flags[cls] == 2)
    if (lane == 0) flags[cls] = 2;
    return;
  }
  if (lane == 0) n_keep[cls] = kept;
  if (MPN_ABLATE(trace != nullptr) && blockIdx.x == 0 && lane == 0) {  // totals of class 0: [kernel cycles, cycles inside simulate(), simulate() calls, batches]
    trace[63 * 8 + 4] = __builtin_amdgcn_s_memtime() - tr_t0; trace[63 * 8 + 5] = tr_sim; trace[63 * 8 + 6] = (unsigned long long)tr_calls; trace[63 * 8 + 7] = (unsigned long long)bid;
  }
}


// Tier 2: classes with bit-equal scores.  The reference's winner among equal scores depends on where
// its array swap (nms.c:83-85) has moved boxes, so the array is emulated exactly — but on the bitmask,
// not by re-sweeping IoUs: ranks (sorted order) index `alive` and the suppression rows; SLOTS (positions
// in the reference's array) index `occ`.  Per pick: first alive rank (ballot/ctz) -> among its alive
// equal-score run take the smallest slot -> head = first occupied slot holding an alive box (lazy
// deletion) -> the head inherits the pick's slot -> alive &= ~row[pick].  One wave per class does the
// serial part; the block's other waves only help stage the mask into LDS.

template <bool LDSMASK>
__global__ __launch_bounds__(256) void nms_tie_kernel(const float4 *__restrict__ sbox, const float *__restrict__ sscore,
                                                      const int *__restrict__ sidx, const int *__restrict__ n_sel,
                                                      const int *__restrict__ flags, const int *__restrict__ counts,
                                                      int m_stride, int w64, const unsigned long long *__restrict__ mask,
                                                      float *__restrict__ keep, int *__restrict__ keep_idx,
                                                      int *__restrict__ n_keep, int m_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
  const int cls = blockIdx.x;
  if (flags[cls] != 1) return;
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  if (LDSMASK != (m <= kTieLdsMask)) return;  // the other instantiation handles this class
  const int n = n_sel[cls];
  const int W = (m + 63) >> 6;
  unsigned long long *mlds = lds64;                                                    // [n][W] when LDSMASK
  int *pos_r = reinterpret_cast<int *>(lds64 + (LDSMASK ? (size_t)kTieLdsMask * (kTieLdsMask / 64) : 0));
  int *box_at = pos_r + m_cap;
  int *gend = box_at + m_cap;
  int *kept_list = gend + m_cap;
  const float4 *b4 = sbox + (size_t)cls * m_stride;
  const float *sc = sscore + (size_t)cls * m_stride;
  const int *si = sidx + (size_t)cls * m_stride;
  const unsigned long long *mk = mask + (size_t)cls * m_stride * w64;
  const int tid = threadIdx.x;
  float *sc_l = reinterpret_cast<float *>(kept_list + m_cap);                 // scores in rank order
  unsigned long long *tiew = reinterpret_cast<unsigned long long *>(sc_l + m_cap);  // [64] tie-next words
  if (LDSMASK) {  // stage the class's mask rows: 8 independent 8-byte loads in flight per thread
    const int total = n * W;
    for (int t0 = 0; t0 < total; t0 += 256 * 8) {
      unsigned long long v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = t0 + j * 256 + tid;
        const int r = t / W, w = t - r * W;
        v[j] = t < total ? mk[(size_t)r * w64 + w] : 0ull;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = t0 + j * 256 + tid;
        if (t < total) mlds[t] = v[j];
      }
    }
  }
  for (int r = tid; r < m; r += blockDim.x) {
    const int x = si[r];
    pos_r[r] = x;
    box_at[x] = r;
    sc_l[r] = sc[r];
  }
  if (tid < 64) tiew[tid] = 0ull;
  __syncthreads();
  for (int r0 = 0; r0 < m; r0 += blockDim.x) {
    const int r = r0 + tid;
    // bit r of the tie words: rank r+1 exists, both are pickable (< n) and carry the same score
    const bool tie = (r + 1 < n) && (sc_l[r] == sc_l[r + 1]);
    const unsigned long long bal = __ballot(tie);
    if ((tid & 63) == 0 && r < m) tiew[r >> 6] = bal;
    if (r < m) {
      const bool start = (r == 0) || !(r < n && sc_l[r - 1] == sc_l[r]);
      if (start) {  // the first element of each equal-score run fills in the run's last rank
        int e = r;
        while (e + 1 < n && sc_l[e + 1] == sc_l[r]) ++e;
        for (int q = r; q <= e; ++q) gend[q] = e;
      }
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  const int lane = tid;
  float *kout = keep + (size_t)cls * m_stride * 5;
  int *kidx = keep_idx ? keep_idx + (size_t)cls * m_stride : nullptr;
  unsigned long long alive = 0ull;
  if (lane < W) { int nv = min(64, m - lane * 64); alive = nv == 64 ? ~0ull : ((1ull << nv) - 1ull); }
  unsigned long long occ = alive;  // slots 0..m-1 all occupied
  const unsigned long long tiebits = tiew[lane];
  int kept = 0;
  for (;;) {
    const unsigned long long bal = __ballot(alive != 0ull);
    if (!bal) break;
    const int L = __builtin_ctzll(bal);
    const int r0 = (L << 6) + __builtin_ctzll(readlane64(alive, L));
    if (r0 >= n) break;  // only boxes with score <= -1e7 are left: the reference would never pick them
    int b = r0;
    int e = r0;
    // r0 is the first alive rank of its run; the run continues past r0 iff r0's tie bit is set
    if ((readlane64(tiebits, r0 >> 6) >> (r0 & 63)) & 1ull) e = __builtin_amdgcn_readfirstlane(gend[r0]);
    if (e > r0) {  // equal-score run: the reference takes the one sitting first in its array
      int bp = 0x7fffffff, br = -1;
      for (int base = r0; base <= e; base += 64) {
        const int r = base + lane;
        bool ok = r <= e;
        const unsigned long long aw = shfl64(alive, (ok ? r : r0) >> 6);
        ok = ok && ((aw >> (r & 63)) & 1ull);
        const int p = ok ? pos_r[r] : 0x7fffffff;
        if (p < bp) { bp = p; br = r; }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const int op = __shfl_xor(bp, off), orr = __shfl_xor(br, off);
        if (op < bp) { bp = op; br = orr; }
      }
      b = __builtin_amdgcn_readfirstlane(br);
    }
    const int sb = __builtin_amdgcn_readfirstlane(pos_r[b]);
    // head of the reference's array = first occupied slot whose box is still alive (lazy deletion)
    int hs, f;
    for (;;) {
      const unsigned long long ob = __ballot(occ != 0ull);
      const int L2 = __builtin_ctzll(ob);
      const int bit = __builtin_ctzll(readlane64(occ, L2));
      hs = (L2 << 6) + bit;
      f = __builtin_amdgcn_readfirstlane(box_at[hs]);
      if ((readlane64(alive, f >> 6) >> (f & 63)) & 1ull) break;
      if (lane == L2) occ &= ~(1ull << bit);
    }
    if (f != b) {  // nms.c:83-85: boxes[0] <-> boxes[best]
      if (lane == (hs >> 6)) occ &= ~(1ull << (hs & 63));
      box_at[sb] = f;
      pos_r[f] = sb;
    } else {
      if (lane == (sb >> 6)) occ &= ~(1ull << (sb & 63));
    }
    kept_list[kept] = b;  // emitted after the loop: no global-memory latency on the serial chain
    ++kept;
    if (lane == (b >> 6)) alive &= ~(1ull << (b & 63));
    unsigned long long row = 0ull;
    if (lane < W) row = LDSMASK ? mlds[(size_t)b * W + lane] : mk[(size_t)b * w64 + lane];
    alive &= ~row;
  }
  for (int k = lane; k < kept; k += kWave) {
    const int b = kept_list[k];
    const float4 bx = b4[b];
    float *q = kout + (size_t)k * 5;
    q[0] = bx.x; q[1] = bx.y; q[2] = bx.z; q[3] = bx.w; q[4] = sc[b];
    if (kidx) kidx[k] = si[b];
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

// bbox_vote for every class of an image at once (Tester_FRCNN.lua:118-124): class c votes its kept boxes against its
// own scored boxes; the voting weights are score^pow (opt.test_bbox_voting_score_pow: scores:pow(p) on a clone of the
// scored boxes, Tester_FRCNN.lua:119-121), applied ONCE per scored box while its tile is staged into LDS and evaluated as
// THFloatTensor_pow does — C pow on the score promoted to double, rounded to float once; pow == 1 skips it, so the
// arithmetic stays bit-identical to nms.c.  Same one-lane-per-kept-box sequential accumulation as bbox_vote_kernel.
__global__ __launch_bounds__(64) void bbox_vote_batched_kernel(const float *__restrict__ keep, const int *__restrict__ n_keep,
                                                               const float *__restrict__ scored, const int *__restrict__ counts,
                                                               int m_stride, float thr, float score_pow, float *__restrict__ res) {
  constexpr int TILE = 256;
  __shared__ float t[TILE * 5];
  const int cls = blockIdx.y;
  const int nk = min(n_keep[cls], m_stride), m = min(counts ? counts[cls] : m_stride, m_stride);
  if (blockIdx.x * kWave >= nk) return;
  const float *kb = keep + (size_t)cls * m_stride * 5;
  const float *sb = scored + (size_t)cls * m_stride * 5;
  float *rb = res + (size_t)cls * m_stride * 5;
  const int i = blockIdx.x * kWave + threadIdx.x;
  const bool act = i < nk;
  float nx1 = 0, ny1 = 0, nx2 = 0, ny2 = 0, ns = 0;
  if (act) { nx1 = kb[5 * i]; ny1 = kb[5 * i + 1]; nx2 = kb[5 * i + 2]; ny2 = kb[5 * i + 3]; ns = kb[5 * i + 4]; }
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
  for (int j0 = 0; j0 < m; j0 += TILE) {
    const int cnt = min(TILE, m - j0);
    __syncthreads();
    for (int q = threadIdx.x; q < cnt * 5; q += kWave) {
      float v = sb[(size_t)j0 * 5 + q];
      if (score_pow != 1.0f && q % 5 == 4) v = (float)pow((double)v, (double)score_pow);
      t[q] = v;
    }
    __syncthreads();
    if (act) {
      for (int j = 0; j < cnt; ++j) {
        const float sx1 = t[5 * j], sy1 = t[5 * j + 1], sx2 = t[5 * j + 2], sy2 = t[5 * j + 3];
        const float ss = t[5 * j + 4];
        const float ov = iou_plus1(sx1, sy1, sx2, sy2, nx1, ny1, nx2, ny2);
        if (ov > thr) { a0 += sx1 * ss; a1 += sy1 * ss; a2 += sx2 * ss; a3 += sy2 * ss; a4 += ss; }
      }
    }
  }
  if (act) {
    rb[5 * i] = a0 / a4; rb[5 * i + 1] = a1 / a4; rb[5 * i + 2] = a2 / a4; rb[5 * i + 3] = a3 / a4;
    rb[5 * i + 4] = ns;
  }
}

__global__ void boxoverlap_kernel(const float *__restrict__ a, int n, float bx1, float by1, float bx2, float by2,
                                  float *__restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = iou_plus1(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3], bx1, by1, bx2, by2);
}

}  // namespace mpn

using namespace mpn;


// =================================================================================================
// Fused NMS for class tables of up to kFusedMax rows (round 5): ONE launch instead of the sort -> mask -> tie -> scan -> wave chain.
//
// grid = (S slices, n_cls classes), 1 block per CU (the greedy phase keeps the class's whole suppression mask in LDS):
//   1. every block of a class sorts the class's keys (score desc, index asc): one key per thread, a bitonic network whose strides below 64
//      are wave shuffles and whose larger strides go through LDS — the S copies of the sort run in parallel and give the same order,
//      which is what lets the mask be split without a second launch;
//   2. slice s computes rows s, s + S, ... of the suppression mask — one wavefront per row, two 64-bit words per trip: lane j evaluates
//      IoU(rank i, rank 64 w + j) exactly as nms.c:14-41 (the row box is an LDS broadcast, the 64 column boxes one conflict-free
//      ds_read_b128 each), a ballot is the word — and writes them THROUGH to memory (device-scope stores: nothing is left dirty in
//      this XCD's L2, so no cache write-back stands between the slices and the block that consumes them);
//   3. the LAST block of the class to finish (device-scope counter) pulls the mask into LDS and one wavefront runs the greedy selection:
//        tie-free class: picks follow the rank order; a chunk of 64 ranks is resolved as a 64-lane fixpoint on its diagonal words
//          (lane j is kept iff it is alive and no KEPT lower lane overlaps it: a handful of ballots), then the kept rows are OR-ed
//          into the per-lane alive words (LDS reads, four in flight);
//        class with bit-equal scores: the reference's pick depends on its array history (nms.c:74-98: FIRST maximum in array order;
//          the old first element takes the picked box's slot; survivors keep their order).  Simulated exactly with two bitsets kept in
//          registers — alive by RANK and occupied by POSITION (slot in the reference's array; dead occupants are dropped lazily, when
//          they surface as the head) — plus pos[rank] / owner[slot] in LDS: a run's pick is its alive member with the smallest slot
//          (the run's slots are cached in registers while the run lasts), the head is the first occupied slot whose owner is alive.
//          One LDS round trip per pick.  (Python model against the compiled nms.c: tools/models/nms_fused_model.py.)
//   4. the whole block writes the kept rows / source indices in pick order.
// A NaN score is never picked by nms.c:77's '>' : such rows sort with the unpickable ones (scores <= -1e7, nms.c:75).
constexpr int kFusedMax = 1024;         // what the kernel can take (its mask must fit one CU's LDS)
constexpr int kFusedDispatchMax = 384;  // ... and what it is sent when the call runs UNDER other work (nms_batched_core)
#ifdef MPN_DEBUG_HOOKS
__device__ unsigned long long g_fused_trace[16];  // s_memtime stamps of class 0's LAST block at the phase boundaries (tools/nms_fused_trace.py)
#define FUSED_STAMP(i) do { if (cls == 0 && tid == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_fused_wall[4096 * 2];  // wall_clock64 (100 MHz, one counter for the whole GPU) at every block's entry and exit
#define FUSED_WALL(k) do { if (tid == 0 && blockIdx.y * gridDim.x + blockIdx.x < 4096) g_fused_wall[(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (k)] = wall_clock64(); } while (0)
__device__ unsigned long long g_fused_sim[8];  // class 0: cycles inside the replay's simulate(), calls, batches
#define FUSED_SIM_T0() const unsigned long long sim_t0 = __builtin_amdgcn_s_memtime(); const int bid0 = bid
#define FUSED_SIM_T1() do { if (blockIdx.y == 0 && lane == 0) { g_fused_sim[0] += __builtin_amdgcn_s_memtime() - sim_t0; g_fused_sim[1] += 1; g_fused_sim[2] += (unsigned long long)(bid - bid0); } } while (0)
#define FUSED_SIM_ACC(i, t_from) do { if (blockIdx.y == 0 && lane == 0) g_fused_sim[i] += __builtin_amdgcn_s_memtime() - (t_from); } while (0)
#define FUSED_SIM_MARK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FUSED_SIM_ACC(i, t_from) do { } while (0)
#define FUSED_SIM_MARK(v) do { } while (0)
#define FUSED_WALL(k) do { } while (0)
#define FUSED_SIM_T0() do { } while (0)
#define FUSED_SIM_T1() do { } while (0)
#define FUSED_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
  return v;
}

// The chunked scan with the LAZY position replay (nms_scan_kernel's flag-3 algorithm, see the comment above it) on the fused kernel's
// LDS-resident mask: one wavefront of the class's last block, the mask rows of ALL boxes full and symmetric (fw = 0), every row read an LDS
// read.  Returns the number of picks (their ranks in klist[0 ..), or -1 if a progress bound was hit (the caller then runs the pick-by-pick
// path, which needs no bound).  n = pickable ranks, m = rows; tiew[w] bit r: ranks r and r + 1 are pickable and carry the same score.
// lds16: pos | occ | rnd | klist | mv (m_cap shorts each), then 384 bytes of round-space scratch (simulate()).
// Measured (tools/nms_fused_trace.py, 1000 rows, the last two kept boxes tied = 651 rounds to replay): 41 batches, ~120 k cycles, a third
// of it the 16 death_of() calls (64 lanes read 64 different mask rows: the rows start on the same LDS banks).
#define FUSED_WAVE_SYNC() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")  /* one wavefront: LDS is in order, this pins the compiler */
__device__ __forceinline__ int fused_replay_select(const unsigned long long *LM, const int lw, const int W, const int m, const int n,
                                                   const unsigned long long *tiew, short *lds16, const int m_cap, const unsigned short *sid,
                                                   const int lane, const bool give_up_midway) {
  typedef unsigned long long u64;
  bool failed = false;
  short *pos = lds16, *occ = lds16 + m_cap, *rnd = lds16 + 2 * m_cap, *klist = lds16 + 3 * m_cap, *mv = lds16 + 4 * m_cap;
  u64 removed = 0ull, keptw = 0ull;  // lane l: ranks 64 l .. 64 l + 63 (m <= 1024: one word per lane)
  for (int r = lane; r < m; r += kWave) {
    const int x = sid[r];
    pos[r] = (short)x; occ[x] = (short)r; rnd[r] = 0; mv[r] = 0;
  }
  FUSED_WAVE_SYNC();
  auto word_of = [&](int w) -> u64 { return readlane64(removed, w & 63); };
  auto kept_word = [&](int w) -> u64 { return readlane64(keptw, w & 63); };
  auto removed_bit = [&](int r) -> bool {  // per-lane rank; every lane of the wave must call it
    const u64 rw = shfl64(removed, (r >> 6) & 63);
    return (rw >> (r & 63)) & 1ull;
  };
  int sim_done = 0, hp = 0, bid = 0;
  const int nw_kept = (n + 63) >> 6;
  auto death_of = [&](int f) -> int {
    const int fs = f < 0 ? 0 : f;
    const int rk = (int)rnd[fs];
    const bool rem = removed_bit(fs);
    int d = f < 0 ? -1 : (rk > 0 ? rk : (rem ? 0 : kForever));
    const bool need = d == 0;  // suppressed: by the kept rank picked first among those that overlap it
    if (__ballot(need)) {
      u64 x[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) x[q] = (need && q < nw_kept) ? LM[(fs << lw) + q] : 0ull;
      int bb = -1;  // the first kept RANK that overlaps fs: branch-free over the words (a divergent body per word cost 4 200 cycles a call)
#pragma unroll
      for (int q = 15; q >= 0; --q) {
        if (q < nw_kept) {  // wave-uniform
          const u64 hit = x[q] & kept_word(q);
          if (hit) bb = q * 64 + __builtin_ctzll(hit);
        }
      }
      const bool found = need && bb >= 0;
      const int b0 = found ? bb : 0;
      int best = (int)rnd[b0];
      const bool in_run = found && ((tiew[b0 >> 6] >> (b0 & 63)) & 1ull);
      if (__ballot(in_run)) {  // rare: the rest of that rank's equal-score run (its picks may be out of rank order)
        if (in_run) {
          int b = b0;
          while ((tiew[b >> 6] >> (b & 63)) & 1ull) {
            ++b;
            const u64 rwd = LM[(fs << lw) + (b >> 6)];
            const int rb = (int)rnd[b];
            if (((rwd >> (b & 63)) & 1ull) && rb > 0 && rb < best) best = rb;
          }
        }
      }
      if (found) d = best;
    }
    return d;
  };
  // Replay of rounds sim_done + 1 .. t1 (all picked: klist / rnd hold them), one 64-slot window per batch.  Unlike nms_scan_kernel's
  // replay, a move that lands INSIDE the window does not end the batch (at 1000 rows that cut a batch every 6.5 rounds: 100 batches of
  // ~1 750 cycles): the window's rounds are laid out in "round space" first — lane j: the pick of round t0 + j, its slot, and through
  // inb[] the window lane that slot is, i.e. where round j's head will land — and the 64-lane fixpoint then runs over (taken, occupant,
  // death) together: the head of round j is the j-th taken lane (Rr[j], written by that lane), lane x's occupant is the head of the round
  // landing on it if that head sits below x.  Lanes settle from the bottom up.  Only a pick that was itself moved earlier in the batch
  // (its slot in pos[] is stale) still ends one.  (tools/models/nms_lazy_replay_model.py: simulate_v2, against the compiled nms.c.)
  int *Rr = reinterpret_cast<int *>(lds16 + 5 * m_cap);  // [64] lane << 22 | rank << 12 | min(death, 4095)
  short *inb = lds16 + 5 * m_cap + 128;                  // [64]
  auto simulate = [&](int t1) {
    int loaded = -64, f0 = -1, d0 = -1;
    const int glim = 2 * m_cap + 1024;
    const u64 below = (1ull << lane) - 1ull;
    for (int guard = 0; sim_done < t1; ++guard) {
      if (guard >= glim) { failed = true; break; }
      FUSED_WAVE_SYNC();
      const int W0 = hp & ~63, p = hp - W0;
      if (W0 != loaded) {
        FUSED_SIM_MARK(td);
        const int sl = W0 + lane;
        f0 = sl < m ? (int)occ[sl] : -1;
        d0 = death_of(f0);
        if (d0 > 4095) d0 = 4095;  // rounds are <= 1024: "alive at round t" tests keep their answers
        loaded = W0;
        FUSED_SIM_ACC(3, td);
      }
      FUSED_SIM_MARK(tp);
      const int t0 = sim_done + 1;
      const int tj = t0 + lane;
      const int pk = tj <= t1 ? (int)klist[tj - 1] : -1;
      const int sbj = pk >= 0 ? (int)pos[pk] : -1;
      inb[lane] = (short)-1;
      FUSED_WAVE_SYNC();
      if (sbj >= W0 && sbj < W0 + 64) inb[sbj - W0] = (short)lane;  // distinct boxes sit in distinct slots
      FUSED_WAVE_SYNC();
      const int jr = (int)inb[lane];
      FUSED_SIM_ACC(4, tp);
      FUSED_SIM_MARK(tf);
      u64 taken = 0ull;
      int f = f0, d = d0;
      bool settled = false;
      for (int it = 0; it < 70; ++it) {
        const int k = __popcll(taken & below), cnt = __popcll(taken);
        if ((taken >> lane) & 1ull) Rr[k] = (lane << 22) | (f << 12) | d;
        FUSED_WAVE_SYNC();
        int nf = f0, nd = d0;
        if (jr >= 0 && jr < cnt) {
          const int v = Rr[jr];
          if ((v >> 22) < lane) { nf = (v >> 12) & 1023; nd = v & 4095; }
        }
        FUSED_WAVE_SYNC();
        const int rd = t0 + k;
        const u64 nt = __ballot(lane >= p && nf >= 0 && nd >= rd && rd <= t1);
        const u64 chg = __ballot(nf != f || nd != d);
        f = nf; d = nd;
        if (nt == taken && !chg) { settled = true; break; }
        taken = nt;
      }
      if (!settled) { failed = true; break; }
      FUSED_SIM_ACC(5, tf);
      FUSED_SIM_MARK(tc);
      const int cnt = __popcll(taken);
      if (cnt == 0) { hp = W0 + 64; if (hp >= m) { failed = true; break; } continue; }
      const bool live = lane < cnt;                       // round space: round t0 + lane was played in this window
      const int hv = live ? Rr[lane] : 0;
      const int hl = hv >> 22, hf = (hv >> 12) & 1023;
      const bool move = live && hf != pk;
      ++bid;
      if (move) mv[hf] = (short)bid;
      FUSED_WAVE_SYNC();
      const u64 h1m = __ballot(live && mv[live ? pk : 0] == (short)bid);
      const int cut = h1m ? __builtin_ctzll(h1m) : cnt;   // >= 1: nothing has been moved before the batch's first round
      if (cut <= 0) { failed = true; break; }
      if (lane < cut && move) { occ[sbj] = (short)hf; pos[hf] = (short)sbj; }
      sim_done += cut;
      if (jr >= 0 && jr < cut) { f0 = f; d0 = d; }        // the window registers follow the committed landings
      hp = (cut == cnt && sim_done < t1) ? W0 + 64        // every lane above the last head is dead on arrival
                                         : W0 + __builtin_amdgcn_readlane(hl, cut - 1) + 1;
      FUSED_SIM_ACC(6, tc);
    }
    FUSED_WAVE_SYNC();
  };
  int kept = 0, sim_run_e = -1;
  const int nchunks = (n + 63) >> 6;
  for (int c = 0; c < nchunks; ++c) {
    if (give_up_midway && c == (nchunks >> 1)) failed = true;  // test hook (replay_on == 2): the caller's redo from a half-used state
    if (failed) break;
    const int base = c << 6;
    const u64 diag = (base + lane < n) ? LM[((base + lane) << lw) + c] : 0ull;
    const int nv = min(64, n - base);
    const u64 valid = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
    const int plim = 2 * m_cap + 130;
    for (int pass = 0;; ++pass) {
      if (pass >= plim || failed) { failed = true; break; }
      const u64 rem_c = word_of(c);
      const u64 alive_all = ~rem_c & valid;
      if (!alive_all) break;
      const u64 tcw = tiew[c];
      const u64 tiesel = alive_all & tcw;
      const int first = __builtin_ctzll(alive_all);
      if (tiesel && __builtin_ctzll(tiesel) == first) {
        // ---- one pick by the exact rule: among the alive members of r0's equal-score run, the one sitting first in the array
        const int r0 = base + first;
        int e = r0;
        for (;;) {
          const u64 ones = tiew[e >> 6] >> (e & 63);
          const int span = 64 - (e & 63);
          const int cnt = (~ones) ? __builtin_ctzll(~ones) : 64;
          if (cnt < span) { e += cnt; break; }
          e += span;
          if (e >= n) { e = n - 1; break; }
        }
        int n_alive = 0, low = 0x7fffffff;
        for (int rr = r0; rr <= e; rr += kWave) {
          const int r = rr + lane;
          const bool in = r <= e;
          const bool ok = !removed_bit(in ? r : r0) && in;
          n_alive += __popcll(__ballot(ok));
          if (ok && r < low) low = r;
        }
        low = wave_min_i32(low);
        int pick = __builtin_amdgcn_readfirstlane(low);
        if (n_alive >= 2) {  // only now do positions matter: bring the slot model up to date, then take the smallest slot
          // ... once per RUN: while a run is being picked its members' slots cannot change — the head that moves in a round is either a
          // non-member (a lower score sitting earlier in the array) or the pick itself (nms_lazy_replay_model.py, V2)
          if (e != sim_run_e) {
            FUSED_SIM_T0();
            simulate(kept);
            FUSED_SIM_T1();
            sim_run_e = e;
          }
          int bp = 0x7fffffff, br = -1;
          for (int rr = r0; rr <= e; rr += kWave) {
            const int r = rr + lane;
            const bool in = r <= e;
            const bool ok = !removed_bit(in ? r : r0) && in;
            const int pp = ok ? (int)pos[r] : 0x7fffffff;
            if (pp < bp) { bp = pp; br = r; }
          }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            const int op = __shfl_xor(bp, off), orr = __shfl_xor(br, off);
            if (op < bp) { bp = op; br = orr; }
          }
          pick = __builtin_amdgcn_readfirstlane(br);
        }
        if (lane == 0) { klist[kept] = (short)pick; rnd[pick] = (short)(kept + 1); }
        if (lane < W) {
          removed |= LM[(pick << lw) + lane];
          if (lane == (pick >> 6)) { removed |= 1ull << (pick & 63); keptw |= 1ull << (pick & 63); }
        }
        ++kept;
        FUSED_WAVE_SYNC();
        continue;
      }
      // ---- tie-free rule for the alive ranks below `limit`
      const int limit = tiesel ? __builtin_ctzll(tiesel) : 64;
      const u64 lim_mask = limit == 64 ? ~0ull : ((1ull << limit) - 1ull);
      u64 keptmask, diag_acc;
      {
        const u64 alive0 = alive_all & lim_mask;
        const u64 lower = diag & ((1ull << lane) - 1ull);
        const bool alive_l = (alive0 >> lane) & 1ull;
        keptmask = alive0;
        for (int it = 0; it < 66; ++it) {
          const u64 kn = __ballot(alive_l && !(lower & keptmask));
          if (kn == keptmask) break;
          keptmask = kn;
        }
        diag_acc = __ballot((diag & keptmask) != 0ull);  // every rank of the chunk overlapped by a kept one
      }
      if ((keptmask >> lane) & 1ull) {
        const int o = kept + __popcll(keptmask & ((1ull << lane) - 1ull));
        klist[o] = (short)(base + lane); rnd[base + lane] = (short)(o + 1);
      }
      if (lane > c && lane < W) {  // fold the kept rows into `removed`: four independent ds_reads in flight per trip
        u64 acc = 0ull, k = keptmask;
        const u64 *col = LM + (base << lw) + lane;
        while (k) {
          const int j0 = __builtin_ctzll(k); k &= k - 1ull;
          const int j1 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
          const int j2 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
          const int j3 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
          acc |= (col[j0 << lw] | col[j1 << lw]) | (col[j2 << lw] | col[j3 << lw]);
        }
        removed |= acc;
      }
      if (lane == c) { removed |= (valid & lim_mask) | (diag_acc & valid); keptw |= keptmask; }
      kept += __popcll(keptmask);
      FUSED_WAVE_SYNC();
      if (limit == 64) break;
    }
  }
  return failed ? -1 : kept;
}

__global__ __launch_bounds__(1024) void nms_fused_kernel(const float *__restrict__ scored, const int *__restrict__ counts, int m_stride, float thr,
                                                         float *__restrict__ keep, int *__restrict__ keep_idx, int *__restrict__ n_keep,
                                                         unsigned long long *gmask, unsigned int *cnt, int lw, int mask_bytes, int replay_on, int fence) {
  // lw = log2 of the mask's row pitch in 64-bit words (a power of two >= ceil(m_stride / 64)), in HBM and in LDS; blockDim.x = the sort
  // width = the power of two >= max(64, m_stride)
  typedef unsigned long long u64;
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
  __shared__ int sh_nsel, sh_last, sh_kept, sh_bad;
  const int cls = blockIdx.y, slice = blockIdx.x, S = gridDim.x, tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
#ifdef MPN_DEBUG_HOOKS
  unsigned long long stamp[12] = {};
#endif
  FUSED_STAMP(0);
  FUSED_WALL(0);
  int m = counts ? counts[cls] : m_stride;
  if (m > m_stride) m = m_stride;
  if (m <= 0) {
    if (slice == 0 && tid == 0) n_keep[cls] = 0;
    return;
  }
  const int W = (m + 63) >> 6, cap = W * 64, Wp = 1 << lw, cap_s = nt;
  u64 *LM = reinterpret_cast<u64 *>(fused_lds);                             // sort exchange buffer [nt], later the class's mask [m][Wp]
  float4 *box = reinterpret_cast<float4 *>(fused_lds + mask_bytes);          // sorted boxes [cap_s]; later klist | pos | owner (u16 [cap_s] each)
  unsigned short *sid = reinterpret_cast<unsigned short *>(fused_lds + mask_bytes + (size_t)cap_s * 16);  // source row of rank r
  unsigned short *fw = sid + cap_s;                                          // first mask word of row r that is computed
  u64 *EQ = reinterpret_cast<u64 *>(fw + cap_s);                             // bit r: ranks r and r + 1 carry the same score   [16 words]
  u64 *TW = EQ + 16;                                                         // the same among PICKABLE ranks only (the replay's tie words) [16 words]
  u64 *keys = LM;
  const float *src = scored + (size_t)cls * m_stride * 5;
  if (tid == 0) { sh_nsel = 0; sh_kept = 0; sh_bad = 0; }
  if (tid < 16) EQ[tid] = 0ull;
  // ---- 1. one key per thread
  u64 key = ~0ull;
  if (tid < m) {
    const float sc = src[5 * (size_t)tid + 4];
    unsigned u = __float_as_uint(sc);
    if ((u << 1) == 0u) u = 0u;          // -0.0f and 0.0f are EQUAL for nms.c:77's '>': one key, so that the array order decides between them
    if (sc != sc) u = 0xff800000u;       // a NaN score is never picked (nms.c:77: `NaN > bestS` is false): it sorts with the unpickable rows, as -inf
    key = ((u64)(~nms_f2key(__uint_as_float(u))) << 32) | (unsigned)tid;
  }
  FUSED_STAMP(1);
  for (int k = 2; k <= nt; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      u64 other;
      if (j >= 64) {  // block-uniform
        __syncthreads();
        keys[tid] = key;
        __syncthreads();
        other = keys[tid ^ j];
      } else {
        other = shfl64(key, lane ^ j);
      }
      const bool take_min = ((tid & j) == 0) == ((tid & k) == 0);  // the lower index of an ascending pair, or the upper one of a descending pair
      const bool smaller = other < key;
      key = (smaller == take_min) ? other : key;
    }
  __syncthreads();
  keys[tid] = key;
  __syncthreads();
  FUSED_STAMP(2);
  // ---- sorted boxes, source rows, equal-score bits, the pickable prefix (nms.c:75: scores <= -1e7 are never picked)
  for (int r = tid; r < cap; r += nt) {  // cap and nt are multiples of 64: whole wavefronts
    const bool valid = r < m;
    const u64 k = valid ? keys[r] : ~0ull;
    const int i = (int)(unsigned)k;
    const bool e = (r + 1 < m) && ((unsigned)(keys[r + 1] >> 32) == (unsigned)(k >> 32));
    bool sel = false;
    if (valid) {
      const float *q = src + 5 * (size_t)i;
      box[r] = make_float4(q[0], q[1], q[2], q[3]);
      sid[r] = (unsigned short)i;
      sel = q[4] > -10000000.0f;  // (false for NaN)
    }
    const u64 be = __ballot(e), bs = __ballot(sel);
    if (lane == 0) {
      EQ[r >> 6] = be;
      if (bs) atomicAdd(&sh_nsel, __popcll(bs));
    }
  }
  __syncthreads();
  const int n_sel = sh_nsel;
  if (tid < 16) {  // tied pairs among the PICKABLE ranks (unpickable rows are never picked: their ties never matter)
    const int lim = n_sel - 1 - 64 * tid;  // bits r with r + 1 < n_sel
    const u64 tw = lim <= 0 ? 0ull : (EQ[tid] & (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull)));
    TW[tid] = tw;
    if (tw) atomicAdd(&sh_bad, __popcll(tw));
  }
  __syncthreads();
  // mode 0: no tied pickable pair — picks follow the rank order; mode 3: a few tied pairs — the chunked scan with the lazy position replay
  // (needs the rows of ALL boxes full and symmetric); mode 1: many — the exact rule pick by pick
  const int bad = sh_bad;
  // The replay pays per tied pair (one simulate() call of >= 1 batch + a window's death search: ~5 k cycles), the pick-by-pick path per
  // pick (1 200-2 400 cycles): measured crossover at ~m / 10 pairs (tools/bench_nms.py ties30 / ties100: 1000 rows 207 vs 371 us at 30
  // pairs, 355 vs 387 at 100; 300 rows 140 vs 144 at 30, 339 vs 165 at 100).
  const int few = replay_on > 2 ? replay_on : max(kFewTies, m / 12);  // (test / tuning hook: > 2 = the threshold itself)
  const int mode = bad == 0 ? 0 : (bad <= few && replay_on ? 3 : 1);
  const bool has_ties = mode != 0;
  // first mask word a row needs: picks inside an equal-score run may come in any rank order, so a row must cover its run from the
  // run's first rank; a rank outside any run only ever suppresses later ranks
  for (int r = tid; r < m; r += nt) {
    int rs = mode == 3 ? 0 : r;
    if (mode == 1) {
      while (rs > 0) {
        const int w = (rs - 1) >> 6, b = (rs - 1) & 63;
        const u64 zeros = ~EQ[w] & (b == 63 ? ~0ull : ((1ull << (b + 1)) - 1ull));  // ranks q <= rs - 1 in this word with score[q] != score[q + 1]
        if (zeros) { rs = 64 * w + (63 - __builtin_clzll(zeros)) + 1; break; }
        rs = 64 * w;
      }
    }
    fw[r] = (unsigned short)(rs >> 6);
  }
  __syncthreads();
  FUSED_STAMP(3);
  // ---- 2. this slice's rows of the suppression mask: one wavefront per row, two words per trip (two independent IoU chains per lane)
  {
    u64 *G = gmask + (((size_t)cls * m_stride) << lw);
    for (int i = slice + wave * S; i < m; i += nwaves * S) {
      const float4 a = box[i];
      for (int w = fw[i]; w < W; w += 2) {
        const int c0 = 64 * w + lane, c1 = c0 + 64;
        bool b0 = false, b1 = false;
        if (c0 < m && c0 != i) {
          const float4 c = box[c0];
          b0 = !(iou_plus1(a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w) <= thr);  // nms.c:93 keeps `iou <= threshold`
        }
        if (c1 < m && c1 != i) {
          const float4 c = box[c1];
          b1 = !(iou_plus1(a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w) <= thr);
        }
        const u64 w0 = __ballot(b0), w1 = __ballot(b1);
        if (lane == 0) {
          __hip_atomic_store(G + ((size_t)i << lw) + w, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (w + 1 < W) __hip_atomic_store(G + ((size_t)i << lw) + w + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
  FUSED_STAMP(4);
  // ---- 3. the last block of the class to arrive runs the selection.  The mask words were stored and are loaded with device-scope
  // (write-through / cache-bypassing) accesses; every wave waits for its stores to complete, the barrier collects the waves, and thread 0
  // counts the block in with ONE acquire-release device-scope atomic: the barrier orders the other waves' completed stores
  // before thread 0's release, the last block's acquire orders the counter's value before the barrier that lets its waves read the mask —
  // a formally synchronised hand-off, not one that leans on gfx950's write-through behaviour.  The release costs one buffer_wbl2 per
  // BLOCK (round 5 measured one per WAVE — a fence in every thread — at 20-35 us per block at 1000 rows, queueing behind the other
  // blocks' write-backs; per block it is measured in profiles/r06_nms_fence.txt).  fence == 0 (debug flavour only): round 5's relaxed form.
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) {
    const unsigned old = fence ? __hip_atomic_fetch_add(&cnt[cls], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_fetch_add(&cnt[cls], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (unsigned)(S - 1);
    if (last) __hip_atomic_store(&cnt[cls], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    sh_last = last;
  }
  __syncthreads();
  if (!sh_last) { FUSED_WALL(1); return; }
  FUSED_STAMP(5);
  {
    const u64 *G = gmask + (((size_t)cls * m_stride) << lw);
    const int total = m << lw;
#pragma unroll 4
    for (int idx = tid; idx < total; idx += nt) {
      const int i = idx >> lw, w = idx & (Wp - 1);
      LM[idx] = (w >= (int)fw[i] && w < W) ? __hip_atomic_load(G + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
  }
  unsigned short *klist = reinterpret_cast<unsigned short *>(box), *pos = klist + cap_s, *owner = pos + cap_s;
  if (mode == 1)
    for (int r = tid; r < m; r += nt) { const int x = sid[r]; pos[r] = (unsigned short)x; owner[x] = (unsigned short)r; }
  __syncthreads();
  FUSED_STAMP(6);
  const unsigned short *kl = klist;  // the pick list the output phase reads
  int run_mode = mode;
  if (wave == 0 && mode == 3) {  // the lazy replay on the LDS mask; its arrays (pos | occ | rnd | klist | mv, cap each) live where the boxes were
    short *l16 = reinterpret_cast<short *>(box);
    const int k3 = fused_replay_select(LM, lw, W, m, n_sel, TW, l16, cap, sid, lane, replay_on == 2);
    if (k3 >= 0) {
      if (lane == 0) sh_kept = k3;
      run_mode = -1;  // done
    } else {  // (a progress bound was hit: cannot happen in a correct run) — redo the class pick by pick
      for (int r = lane; r < m; r += kWave) { const int x = sid[r]; pos[r] = (unsigned short)x; owner[x] = (unsigned short)r; }
      FUSED_WAVE_SYNC();
      run_mode = 1;
    }
  }
  if (mode == 3) kl = reinterpret_cast<const unsigned short *>(box) + 3 * cap;
  if (wave == 0 && run_mode >= 0) {
    int kept = 0;
    u64 aw = 0ull;  // lane l: alive ranks 64 l .. 64 l + 63
    {
      const int lim = has_ties ? m : n_sel;  // tie-free: the unpickable suffix never matters
      const int lo = 64 * lane;
      if (lo < lim) aw = lim - lo >= 64 ? ~0ull : ((1ull << (lim - lo)) - 1ull);
    }
    if (run_mode == 0) {
      for (int c = 0; c < W; ++c) {
        const u64 cur = readlane64(aw, c);
        if (!cur) continue;
        const int r = 64 * c + lane;
        const u64 D = r < m ? LM[(r << lw) + c] : 0ull;       // lane j: the chunk's own columns of rank 64 c + j
        const u64 T = D & ((1ull << lane) - 1ull);              // the lower lanes that overlap me (IoU is symmetric, bit for bit)
        const bool al = (cur >> lane) & 1ull;
        u64 kw = cur;
        for (int it = 0; it < 66; ++it) {  // lane j is final once the lanes below it are: <= 64 rounds, typically 2-4
          const u64 nk = __ballot(al && !(T & kw));
          if (nk == kw) break;
          kw = nk;
        }
        if ((kw >> lane) & 1ull) klist[kept + __popcll(kw & ((1ull << lane) - 1ull))] = (unsigned short)r;
        kept += __popcll(kw);
        if (lane > c && lane < W) {  // four independent ds_reads in flight per trip
          u64 rem = 0ull, k = kw;
          const u64 *col = LM + ((64 * c) << lw) + lane;
          while (k) {
            const int j0 = __builtin_ctzll(k); k &= k - 1ull;
            const int j1 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
            const int j2 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
            const int j3 = k ? __builtin_ctzll(k) : j0; k &= k - 1ull;
            const u64 v0 = col[j0 << lw], v1 = col[j1 << lw], v2 = col[j2 << lw], v3 = col[j3 << lw];
            rem |= (v0 | v1) | (v2 | v3);
          }
          aw &= ~rem;
        }
      }
    } else {
      const u64 eqw = lane < 16 ? EQ[lane] : 0ull;                                                        // lane l: EQ word l
      u64 apw = 64 * lane >= m ? 0ull : (m - 64 * lane >= 64 ? ~0ull : ((1ull << (m - 64 * lane)) - 1ull)); // lane l: occupied slots 64 l ..
      int run_s = 0, run_e = -1, mp = 0x7fffffff;  // the cached equal-score run [run_s, run_e] (<= 64 members): lane j holds pos[run_s + j]
      for (;;) {
        const u64 nz = __ballot(aw != 0ull);
        if (!nz) break;
        const int w0 = __builtin_ctzll(nz);
        const u64 word0 = readlane64(aw, w0);
        const int r0 = 64 * w0 + __builtin_ctzll(word0);
        if (r0 >= n_sel) break;  // only unpickable rows are left
        // ---- the pick: r0, or — inside an equal-score run — the run's alive member that sits first in the reference's array
        int b = r0, pb = -1;
        const u64 eq0 = readlane64(eqw, w0);
        if ((eq0 >> (r0 & 63)) & 1ull) {
          int w = w0;
          u64 x = ~eq0 & (~0ull << (r0 & 63));
          while (!x) { ++w; x = ~readlane64(eqw, w); }  // bit m - 1 is never set: terminates inside the table
          const int e = 64 * w + __builtin_ctzll(x);    // last rank of the run; its members below r0 are dead
          if (e - r0 < 64) {
            if (e != run_e) {  // a new run: its members' slots into registers (one LDS round trip per run)
              run_s = r0; run_e = e;
              mp = (run_s + lane <= e) ? (int)pos[run_s + lane] : 0x7fffffff;
            }
            const int r = run_s + lane;
            const int wa = run_s >> 6;
            const u64 a0 = readlane64(aw, wa), a1 = readlane64(aw, wa + 1 < 16 ? wa + 1 : 15);
            const bool in = r <= run_e && ((((r >> 6) == wa ? a0 : a1) >> (r & 63)) & 1ull);
            const int p = in ? mp : 0x7fffffff;
            u64 mb = __ballot(in);
            int best = 0x7fffffff;
            if (__popcll(mb) <= 8) {  // a few members: scalar minimum over their lanes
              while (mb) {
                const int l = __builtin_ctzll(mb);
                mb &= mb - 1ull;
                const int q = __builtin_amdgcn_readlane(p, l);
                best = q < best ? q : best;
              }
            } else {
              best = __builtin_amdgcn_readfirstlane(wave_min_i32(p));
            }
            const u64 hit = __ballot(in && p == best);
            b = run_s + __builtin_ctzll(hit);
            pb = best;
          } else {  // a run of more than 64 members (saturated scores): its alive members' slots straight from LDS
            int best = 0x7fffffff;
            for (int ww = w0; ww <= (e >> 6); ++ww) {
              const u64 a = readlane64(aw, ww);
              const int r = 64 * ww + lane;
              const bool in = r >= r0 && r <= e && ((a >> lane) & 1ull);
              const int p = in ? (int)pos[r] : 0x7fffffff;
              best = p < best ? p : best;
            }
            best = __builtin_amdgcn_readfirstlane(wave_min_i32(best));
            b = __builtin_amdgcn_readfirstlane((int)owner[best]);
            pb = best;
          }
        }
        // ---- one LDS round trip: the pick's mask row, its slot, and the owner of the first occupied slot (the head candidate)
        const u64 row = lane < W ? LM[(b << lw) + lane] : 0ull;
        int pbv = pb < 0 ? (int)pos[b] : pb;
        int pw, pf, f;
        u64 pword;
        for (;;) {  // nms.c:83-85's boxes[0]: the first occupied slot whose owner is still alive (dead owners are dropped here, lazily)
          const u64 nzp = __ballot(apw != 0ull);
          pw = __builtin_ctzll(nzp);
          pword = readlane64(apw, pw);
          pf = 64 * pw + __builtin_ctzll(pword);
          f = __builtin_amdgcn_readfirstlane((int)owner[pf]);
          if ((readlane64(aw, f >> 6) >> (f & 63)) & 1ull) break;
          if (lane == pw) apw &= ~(1ull << (pf & 63));
        }
        pbv = __builtin_amdgcn_readfirstlane(pbv);
        // nms.c:83-85: boxes[0] <-> boxes[best] — the head takes the pick's slot (which stays occupied), the pick leaves the array
        if (lane == pw) apw &= ~(1ull << (pf & 63));
        if (pf != pbv) {
          if (lane == 0) { owner[pbv] = (unsigned short)f; pos[f] = (unsigned short)pbv; }
          if (f >= run_s && f <= run_e && lane == f - run_s) mp = pbv;  // the head is a member of the cached run
        }
        if (lane == 0) klist[kept] = (unsigned short)b;
        ++kept;
        // nms.c:91-98: the survivors keep `iou <= threshold`
        if (lane == (b >> 6)) aw &= ~(1ull << (b & 63));
        aw &= ~row;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // lane 0's slot update before the next round's reads (LDS is in order per wave)
      }
    }
    if (lane == 0) { sh_kept = kept; if (mode == 3) sh_last = 2; }  // (2: the replay gave up, the picks are in the kernel's own list)
  }
  __syncthreads();
  if (sh_last == 2) kl = klist;
  FUSED_STAMP(7);
  // ---- 4. kept rows in pick order
  const int K = sh_kept;
  float *kout = keep + (size_t)cls * m_stride * 5;
  for (int t = tid; t < K * 5; t += nt) {
    const int k = t / 5, fcol = t - 5 * k;
    kout[t] = src[5 * (size_t)sid[kl[k]] + fcol];
  }
  if (keep_idx) {
    int *kidx = keep_idx + (size_t)cls * m_stride;
    for (int k = tid; k < K; k += nt) kidx[k] = (int)sid[kl[k]];
  }
  if (tid == 0) n_keep[cls] = K;
#ifdef MPN_DEBUG_HOOKS
  FUSED_STAMP(8);
  FUSED_WALL(1);
  if (cls == 0 && tid == 0) {
    for (int i = 0; i < 9; ++i) g_fused_trace[i] = stamp[i];
    g_fused_trace[9] = (unsigned long long)K; g_fused_trace[10] = (unsigned long long)sh_bad;
    for (int i = 0; i < 3; ++i) g_fused_trace[11 + i] = g_fused_sim[i];
    for (int i = 0; i < 8; ++i) { g_fused_wall[8000 + i] = g_fused_sim[i]; g_fused_sim[i] = 0ull; }
  }
#endif
}

MPN_KNOB(int, g_nms_force_exact, 0);  // test hook: 1 = always the exact IoU-sweep kernel, 2 = always the tie (slot-emulation) kernel, 3 = always the replaying scan
MPN_KNOB(unsigned long long *, g_nms_trace, nullptr);
MPN_KNOB(int, g_nms_guard_limit, 0);  // test hook (mpn_debug_set_nms_guard_limit): bound of the replaying scan's progress loops (0 = the real one)
MPN_KNOB(int, g_nms_fused_replay, 1);  // test hook: 0 = classes with a few tied pairs take the fused kernel's pick-by-pick path instead of the lazy replay; 2 = the replay gives up half way (its progress bounds cannot be reached otherwise) and the class is redone pick by pick
MPN_KNOB(int, g_nms_fused_slices, 0);  // test / timing hook: mask slices per class of the fused kernel (0 = fill the GPU once)
MPN_KNOB(int, g_nms_fused_fence, 1);  // timing hook (mpn_debug_set_nms_fused_fence): 0 = the fused kernel's block hand-off with relaxed atomics (round 5's form)
MPN_KNOB(int, g_nms_fused, 1);  // test hook (mpn_debug_set_nms_fused): 0 = the launch chain at every size; 2 = the fused kernel for every table of <= kFusedMax rows
#ifdef MPN_DEBUG_HOOKS
extern "C" void mpn_debug_set_nms_force_exact(int v) { g_nms_force_exact = v; }
extern "C" void mpn_debug_set_nms_trace(void *p) { g_nms_trace = static_cast<unsigned long long *>(p); }
extern "C" void mpn_debug_set_nms_guard_limit(int v) { g_nms_guard_limit = v; }
extern "C" void mpn_debug_set_nms_fused(int v) { g_nms_fused = v; }
extern "C" void mpn_debug_set_nms_fused_fence(int v) { g_nms_fused_fence = v; }
extern "C" void mpn_debug_set_nms_fused_slices(int v) { g_nms_fused_slices = v; }
extern "C" void mpn_debug_set_nms_fused_replay(int v) { g_nms_fused_replay = v; }
extern "C" int mpn_debug_get_nms_fused_wall(unsigned long long *h_out, int n_blocks) {  // [n_blocks][2]: entry, exit (10 ns ticks)
  MPN_CHECK_ARG(h_out && n_blocks >= 0 && n_blocks <= 4096);
  MPN_CHECK_HIP(hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_fused_wall), (size_t)n_blocks * 2 * sizeof(unsigned long long)));
  return MPN_OK;
}
extern "C" int mpn_debug_get_nms_fused_trace(unsigned long long *h_out16) {
  MPN_CHECK_ARG(h_out16);
  MPN_CHECK_HIP(hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(g_fused_trace), 16 * sizeof(unsigned long long)));
  return MPN_OK;
}
#endif

static int nms_batched_core(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep, int *d_keep_idx,
                            int *d_n_keep, void *stream, bool under_other_work);
extern "C" int mpn_nms_batched(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr,
                               float *d_keep, int *d_keep_idx, int *d_n_keep, void *stream) {
  return nms_batched_core(d_scored, d_counts, n_cls, m_stride, thr, d_keep, d_keep_idx, d_n_keep, stream, false);
}
namespace mpn {
// the pipelined forms' tail: NMS on a side stream UNDER the next image's trunk (mpn_internal.h)
int nms_batched_under_trunk(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep, int *d_keep_idx,
                            int *d_n_keep, hipStream_t stream) {
  return nms_batched_core(d_scored, d_counts, n_cls, m_stride, thr, d_keep, d_keep_idx, d_n_keep, stream, true);
}
}  // namespace mpn

static int nms_batched_core(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep, int *d_keep_idx,
                            int *d_n_keep, void *stream, bool under_other_work) {
  MPN_CHECK_ARG(n_cls >= 0 && m_stride >= 0);
  MPN_CHECK_ARG(d_n_keep != nullptr);
  if (n_cls == 0) return MPN_OK;
  if (m_stride == 0) {
    MPN_CHECK_HIP(hipMemsetAsync(d_n_keep, 0, sizeof(int) * n_cls, as_stream(stream)));
    return MPN_OK;
  }
  MPN_CHECK_ARG(d_scored != nullptr && d_keep != nullptr);
  hipStream_t st = as_stream(stream);
  if (m_stride > MPN_NMS_MAX_BOXES) {  // beyond the LDS-resident paths: the exact sweep kernel on HBM-resident working arrays
    const int m_cap = (m_stride + 3) & ~3;
    void *ws = nullptr;
    int rc_ws = scratch_get(SCR_NMS, (size_t)n_cls * 6 * m_cap * sizeof(float), st, &ws);
    if (rc_ws) return rc_ws;
    hipLaunchKernelGGL(nms_wave_kernel, dim3(n_cls), dim3(kWave), 0, st, d_scored, d_counts, m_stride, thr, d_keep, d_keep_idx, d_n_keep, m_cap,
                       (const int *)nullptr, static_cast<float *>(ws));
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  // Dispatch (measured, profiles/r05_nms_paths.txt).  The fused kernel takes every table of <= kFusedMax rows when the call has the GPU to
  // itself (module-level calls, the un-pipelined test_one, the latency mode, libnms.so): one launch, the mask in LDS, tie-free classes 70
  // cycles per pick, classes with a few tied pairs the lazy position replay on LDS rows, classes full of ties 1 700 cycles per pick.  Its
  // blocks hold 150 KB of LDS each, though: UNDER the next image's trunk (the pipelined forms' side stream) 240 of them displace the trunk's
  // one-block-per-CU launches, where the chain's small kernels slip in beside them — there tables above kFusedDispatchMax rows keep the chain
  // (the headline's 1000-row tables: 280.8 k proposals/s with the fused kernel under the trunk, 289-291 k with the chain).
  // (mpn_debug_set_nms_fused: 0 = the chain at every size, 2 = the fused kernel wherever it can run.)
  const int fused_limit = g_nms_fused == 2 ? kFusedMax : (under_other_work ? kFusedDispatchMax : kFusedMax);
  if (m_stride <= fused_limit && g_nms_fused && g_nms_force_exact == 0) {  // one launch: nms_fused_kernel
    const int cap_w = (m_stride + 63) / 64;
    int lw = 0;
    while ((1 << lw) < cap_w) ++lw;
    int nt = 64;
    while (nt < m_stride) nt <<= 1;                              // one sort key per thread
    size_t mask_bytes = ((size_t)m_stride << lw) * 8;            // the mask [m][1 << lw] in LDS ...
    if (mask_bytes < (size_t)nt * 8) mask_bytes = (size_t)nt * 8;  // ... aliasing the sort's exchange buffer
    mask_bytes = (mask_bytes + 15) & ~(size_t)15;
    const size_t lds = mask_bytes + (size_t)nt * 16 + (size_t)nt * 4 + 256;
    int S = g_nms_fused_slices > 0 ? g_nms_fused_slices : 256 / n_cls;  // one block per CU (the mask lives in LDS): at most one wave of blocks over the 256 CUs
    if (S > 16) S = 16;
    if (S > cdiv(m_stride, 32)) S = cdiv(m_stride, 32);
    if (S < 1) S = 1;
    const size_t gm_bytes = ((((size_t)n_cls * m_stride) << lw) * 8 + 255) & ~(size_t)255;
    void *ws = nullptr, *wc = nullptr;
    int rc_ws = scratch_get(SCR_NMS, gm_bytes, st, &ws);
    if (rc_ws == MPN_OK) rc_ws = scratch_get_zeroed(SCR_NMS_CNT, ((size_t)n_cls * sizeof(unsigned) + 4095) & ~(size_t)4095, st, &wc);
    if (rc_ws) return rc_ws;
    { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_fused_kernel), 160 * 1024 - 64); if (rc_attr) return rc_attr; }
    hipLaunchKernelGGL(nms_fused_kernel, dim3(S, n_cls), dim3(nt), lds, st, d_scored, d_counts, m_stride, thr, d_keep, d_keep_idx, d_n_keep,
                       static_cast<unsigned long long *>(ws), static_cast<unsigned int *>(wc), lw, (int)mask_bytes, (int)g_nms_fused_replay, (int)g_nms_fused_fence);
    {
      const hipError_t e_launch = hipGetLastError();
      if (e_launch != hipSuccess) {  // the per-class arrival counters reset themselves only when a launch runs to its end: never leave them in doubt
        (void)hipMemsetAsync(wc, 0, (size_t)n_cls * sizeof(unsigned), st);
        MPN_CHECK_HIP(e_launch);
      }
    }
    return MPN_OK;
  }
  // ---- scratch for the fast path (library-owned, grown on demand, one stream at a time)
  const int w64 = (m_stride + 63) / 64;
  const size_t n_rows = (size_t)n_cls * m_stride;
  const size_t need = n_rows * (sizeof(float4) + sizeof(float) + sizeof(int)) + n_rows * w64 * sizeof(unsigned long long) +
                      (size_t)n_cls * 2 * sizeof(int) + 256 + (size_t)64 * w64 * sizeof(unsigned long long);  // + a chunk of rows the scan may over-read
  char *scratch = nullptr;
  {
    void *ws = nullptr;
    int rc_ws = scratch_get(SCR_NMS, need, st, &ws);
    if (rc_ws) return rc_ws;
    scratch = static_cast<char *>(ws);
  }
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(scratch);
  float4 *sbox = reinterpret_cast<float4 *>(scratch + n_rows * w64 * sizeof(unsigned long long));
  float *sscore = reinterpret_cast<float *>(sbox + n_rows);
  int *sidx = reinterpret_cast<int *>(sscore + n_rows);
  int *n_sel = sidx + n_rows;
  int *flags = n_sel + n_cls;
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_sort_kernel), kSortMax * 8); if (rc_attr) return rc_attr; }
  int n_pad = 64;
  while (n_pad < m_stride && n_pad < kSortMax) n_pad <<= 1;
  const int sort_threads = n_pad / 2 < 64 ? 64 : (n_pad / 2 > 1024 ? 1024 : n_pad / 2);
  {
    hipLaunchKernelGGL(nms_sort_kernel, dim3(n_cls), dim3(sort_threads), (size_t)n_pad * 8, st, d_scored, d_counts, m_stride, sbox,
                       sscore, sidx, n_sel, flags, g_nms_force_exact);
    MPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(w64, cdiv(m_stride, 256), n_cls), dim3(256), 0, st, sbox, n_sel, flags, d_counts,
                       m_stride, w64, thr, mask);
    MPN_CHECK_LAUNCH();
    {  // tie classes (exact slot emulation on the bitmask); both instantiations exit at once when not needed
      const int tcap = (m_stride + 3) & ~3;
      const size_t lds_a = (size_t)kTieLdsMask * (kTieLdsMask / 64) * 8 + (size_t)5 * kTieLdsMask * 4 + 512;
      const size_t lds_b = (size_t)5 * kTieMax * 4 + 512;
      {
        int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_tie_kernel<true>), (int)lds_a);
        if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_tie_kernel<false>), (int)lds_b);
        if (rc_attr) return rc_attr;
      }
      if (m_stride <= kTieMax) {
        const int cap_a = tcap < kTieLdsMask ? tcap : kTieLdsMask;
        hipLaunchKernelGGL(nms_tie_kernel<true>, dim3(n_cls), dim3(256), (size_t)kTieLdsMask * (kTieLdsMask / 64) * 8 + (size_t)5 * cap_a * 4 + 512, st,
                           sbox, sscore, sidx, n_sel, flags, d_counts, m_stride, w64, mask, d_keep, d_keep_idx, d_n_keep, cap_a);
        MPN_CHECK_LAUNCH();
        if (m_stride > kTieLdsMask) {
          hipLaunchKernelGGL(nms_tie_kernel<false>, dim3(n_cls), dim3(256), (size_t)5 * tcap * 4 + 512, st, sbox, sscore, sidx, n_sel, flags,
                             d_counts, m_stride, w64, mask, d_keep, d_keep_idx, d_n_keep, tcap);
          MPN_CHECK_LAUNCH();
        }
      }
    }
    {
      const int scap = (m_stride + 7) & ~7;
      const size_t slds = (size_t)5 * scap * sizeof(short);
      int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<1>), 5 * 4096 * 2);
      if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<2>), 5 * 8192 * 2);
      if (rc_attr) return rc_attr;
      if (w64 <= 64)
        hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(n_cls), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, d_counts, m_stride, w64, mask,
                           d_keep, d_keep_idx, d_n_keep, scap, g_nms_trace, g_nms_guard_limit);
      else
        hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(n_cls), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, d_counts, m_stride, w64, mask,
                           d_keep, d_keep_idx, d_n_keep, scap, g_nms_trace, g_nms_guard_limit);
      MPN_CHECK_LAUNCH();
    }
  }
  int m_cap = (m_stride + 3) & ~3;
  size_t lds = (size_t)m_cap * 6 * sizeof(float);
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_wave_kernel), MPN_NMS_MAX_BOXES * 6 * 4); if (rc_attr) return rc_attr; }
  hipLaunchKernelGGL(nms_wave_kernel, dim3(n_cls), dim3(kWave), lds, st, d_scored, d_counts, m_stride,
                     thr, d_keep, d_keep_idx, d_n_keep, m_cap, flags, (float *)nullptr);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void picks_to_one_based_kernel(int *__restrict__ pick, const int *__restrict__ n_pick, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < min(*n_pick, m)) pick[i] += 1;
}

// ---- utils.nms_dense beyond the LDS sort (m > kSortMax): the reference function has no size limit -----------------------
// Rank by counting: box i's position in (score descending, index ascending) order = the number of boxes whose 64-bit key — the one
// nms_sort_kernel sorts — is smaller; keys are unique (they carry the index), so the ranks are a permutation.  O(m^2) key compares from
// LDS tiles; exactly the order of the LDS bitonic sort.
__global__ __launch_bounds__(256) void dense_rank_scatter_kernel(const float *__restrict__ scored, int m, float4 *__restrict__ sbox, int *__restrict__ sidx) {
  __shared__ unsigned long long tile[1024];
  const int i = blockIdx.x * 256 + threadIdx.x;
  unsigned long long mine = ~0ull;
  if (i < m) mine = ((unsigned long long)(~nms_f2key(scored[5 * (size_t)i + 4])) << 32) | (unsigned)i;
  int rank = 0;
  for (int j0 = 0; j0 < m; j0 += 1024) {
    for (int t = threadIdx.x; t < 1024; t += 256) {
      const int j = j0 + t;
      tile[t] = j < m ? (((unsigned long long)(~nms_f2key(scored[5 * (size_t)j + 4])) << 32) | (unsigned)j) : ~0ull;
    }
    __syncthreads();
    const int n = min(1024, m - j0);
    for (int t = 0; t < n; ++t) rank += tile[t] < mine ? 1 : 0;
    __syncthreads();
  }
  if (i < m) {
    const float *r = scored + 5 * (size_t)i;
    sbox[rank] = make_float4(r[0], r[1], r[2], r[3]);
    sidx[rank] = i;
  }
}

// The walk of utils.lua:416-460 itself on the sorted table, one block: rank i, if still alive, is picked and suppresses every later
// alive rank whose dense overlap with it exceeds `overlap`.  One barrier per PICK (a dead rank writes nothing).  alive: m bytes, all 1.
__global__ __launch_bounds__(1024) void nms_dense_sweep_kernel(const float4 *__restrict__ sbox, const int *__restrict__ sidx, int m, float overlap,
                                                               unsigned char *__restrict__ alive, int *__restrict__ pick, int *__restrict__ n_pick) {
  int np = 0;
  for (int i = 0; i < m; ++i) {
    if (!alive[i]) continue;  // uniform: written before the last barrier
    const float4 c = sbox[i];
    if (threadIdx.x == 0) pick[np] = sidx[i];
    ++np;
    for (int j = i + 1 + (int)threadIdx.x; j < m; j += 1024)
      if (alive[j] && dense_suppresses(c, sbox[j], overlap)) alive[j] = 0;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_pick = np;
}

MPN_KNOB(int, g_nms_dense_sweep, 0);  // test hook (mpn_debug_set_nms_dense_sweep): 1 = always the counting-rank + sweep form
#ifdef MPN_DEBUG_HOOKS
extern "C" void mpn_debug_set_nms_dense_sweep(int v) { g_nms_dense_sweep = v; }
#endif

// utils.nms_dense (utils.lua:402-462; demo.lua:85): "another version of nms that returns indexes instead of new boxes" — sort by
// score (descending), walk the sorted list, a picked box suppresses every box whose area-based IoU with it exceeds `overlap`.
// The pick order is the SORT's (no swap history as in nms.c), so the sort / mask / chunked-scan kernels are reused with the dense
// suppression rule.  torch.sort is TH's quicksort — not stable — and TH is absent from the reference tree: the order among
// bit-equal scores is PARITY UNPINNED and defined here as ascending index.  d_pick [m] receives the 1-based indices
// (the LongTensor the Lua function returns), *d_n_pick their number.
extern "C" int mpn_nms_dense(const float *d_boxes, int m, float overlap, int *d_pick, int *d_n_pick, void *stream) {
  MPN_CHECK_ARG(m >= 0 && d_n_pick != nullptr);
  hipStream_t st = as_stream(stream);
  if (m == 0) { MPN_CHECK_HIP(hipMemsetAsync(d_n_pick, 0, sizeof(int), st)); return MPN_OK; }
  MPN_CHECK_ARG(d_boxes != nullptr && d_pick != nullptr);
  if (m > kSortMax || g_nms_dense_sweep) {  // wider than the LDS sort: rank by counting, then the exact sequential walk (no size limit)
    const size_t need_s = (size_t)m * (sizeof(float4) + sizeof(int) + 1) + 64;
    void *ws_s = nullptr;
    { int rc_ws = scratch_get(SCR_NMS, need_s, st, &ws_s); if (rc_ws) return rc_ws; }
    float4 *sb = static_cast<float4 *>(ws_s);
    int *si = reinterpret_cast<int *>(sb + m);
    unsigned char *alive = reinterpret_cast<unsigned char *>(si + m);
    MPN_CHECK_HIP(hipMemsetAsync(alive, 1, (size_t)m, st));
    hipLaunchKernelGGL(dense_rank_scatter_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, d_boxes, m, sb, si);
    MPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(nms_dense_sweep_kernel, dim3(1), dim3(1024), 0, st, sb, si, m, overlap, alive, d_pick, d_n_pick);
    MPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(picks_to_one_based_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, d_pick, d_n_pick, m);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  const int w64 = (m + 63) / 64;
  const size_t n_rows = (size_t)m;
  const size_t need = n_rows * (sizeof(float4) + sizeof(float) + sizeof(int) + 5 * sizeof(float)) + n_rows * w64 * sizeof(unsigned long long) +
                      2 * sizeof(int) + 256 + (size_t)64 * w64 * sizeof(unsigned long long);
  void *ws = nullptr;
  { int rc_ws = scratch_get(SCR_NMS, need, st, &ws); if (rc_ws) return rc_ws; }
  char *scratch = static_cast<char *>(ws);
  unsigned long long *mask = reinterpret_cast<unsigned long long *>(scratch);
  float4 *sbox = reinterpret_cast<float4 *>(scratch + (((n_rows + 64) * w64 + 1) & ~(size_t)1) * sizeof(unsigned long long));
  float *sscore = reinterpret_cast<float *>(sbox + n_rows);
  int *sidx = reinterpret_cast<int *>(sscore + n_rows);
  float *keep = reinterpret_cast<float *>(sidx + n_rows);
  int *n_sel = reinterpret_cast<int *>(keep + 5 * n_rows);
  int *flags = n_sel + 1;
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_sort_kernel), kSortMax * 8); if (rc_attr) return rc_attr; }
  int n_pad = 64;
  while (n_pad < m) n_pad <<= 1;
  const int sort_threads = n_pad / 2 < 64 ? 64 : (n_pad / 2 > 1024 ? 1024 : n_pad / 2);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(sort_threads), (size_t)n_pad * 8, st, d_boxes, (const int *)nullptr, m, sbox, sscore, sidx, n_sel, flags, 4);
  MPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(w64, cdiv(m, 256), 1), dim3(256), 0, st, sbox, n_sel, flags, (const int *)nullptr, m, w64, overlap, mask);
  MPN_CHECK_LAUNCH();
  const int scap = (m + 7) & ~7;
  const size_t slds = (size_t)5 * scap * sizeof(short);
  int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<1>), 5 * 4096 * 2);
  if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<2>), 5 * 8192 * 2);
  if (rc_attr) return rc_attr;
  if (w64 <= 64)
    hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(1), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, (const int *)nullptr, m, w64, mask, keep, d_pick,
                       d_n_pick, scap, (unsigned long long *)nullptr, 0);
  else
    hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(1), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, (const int *)nullptr, m, w64, mask, keep, d_pick,
                       d_n_pick, scap, (unsigned long long *)nullptr, 0);
  MPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(picks_to_one_based_kernel, dim3(cdiv(m, 256)), dim3(256), 0, st, d_pick, d_n_pick, m);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_nms(const float *d_scored, int m, float thr, float *d_keep, int *d_keep_idx, int *d_n_keep,
                       void *stream) {
  return mpn_nms_batched(d_scored, nullptr, 1, m, thr, d_keep, d_keep_idx, d_n_keep, stream);
}

// The libnms.so drop-in's small tables (Tester_FRCNN.lua:117 hands utils.nms a few hundred rows per class): no device allocation, no
// hipMemcpy — the table is staged in a per-thread pinned, device-mapped buffer which the fused kernel reads and writes directly over the
// host link (6 KB in, <= 7 KB out), so a call is memcpy + one launch + one stream sync + memcpy.  The buffer (45 KB) is allocated once per
// host thread and deliberately never freed: the HIP runtime may already be gone when thread-local destructors run at process exit.
namespace {
struct HostNmsStage { void *pin = nullptr; void *dptr = nullptr; int device = -1; hipStream_t stream = nullptr; int stream_device = -1; };
thread_local HostNmsStage t_host_nms;
constexpr size_t kHostNmsBytes = (size_t)kFusedMax * (5 + 5 + 1) * 4 + 256;
}  // namespace

// The host entries run on a stream of the CALLING THREAD's own: library scratch is kept per (device, stream), so two host threads
// on one device — the reference's worker threads all call utils.nms (Tester_FRCNN.lua:117) — no longer share the NULL stream's NMS scratch,
// whose regrow (sync + free + malloc) by one thread could pull the buffer from under the other thread's launch.  Like the pinned stage, the
// stream is created once per thread and device and never destroyed (the runtime may be gone when thread-local destructors run).
static int host_nms_stream(hipStream_t *out) {
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  HostNmsStage &t = t_host_nms;
  if (!t.stream || t.stream_device != dev) {
    hipStream_t s = nullptr;
    MPN_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    t.stream = s; t.stream_device = dev;
  }
  *out = t.stream;
  return MPN_OK;
}

static int nms_host_small(const float *h_scored, int m, float thr, float *h_keep, int *h_keep_idx, int *n_keep) {
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  hipStream_t ts = nullptr;
  { int rc_s = host_nms_stream(&ts); if (rc_s) return rc_s; }
  HostNmsStage &t = t_host_nms;
  if (!t.pin || t.device != dev) {  // (a thread that switches devices re-stages; the old buffer stays with its device)
    void *p = nullptr, *d = nullptr;
    MPN_CHECK_HIP(hipHostMalloc(&p, kHostNmsBytes, hipHostMallocPortable | hipHostMallocMapped));
    if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(p); set_error("mpn_nms_host: the pinned stage is not device-mapped"); return MPN_EHIP; }
    t.pin = p; t.dptr = d; t.device = dev;
  }
  float *in = static_cast<float *>(t.pin), *out = in + 5 * (size_t)kFusedMax;
  int *idx = reinterpret_cast<int *>(out + 5 * (size_t)kFusedMax), *n = idx + kFusedMax;
  const ptrdiff_t delta = static_cast<char *>(t.dptr) - static_cast<char *>(t.pin);
  auto dp = [&](void *h) { return static_cast<void *>(static_cast<char *>(h) + delta); };
  memcpy(in, h_scored, sizeof(float) * 5 * (size_t)m);
  *n = 0;
  int rc = mpn_nms_batched(static_cast<float *>(dp(in)), nullptr, 1, m, thr, static_cast<float *>(dp(out)), static_cast<int *>(dp(idx)), static_cast<int *>(dp(n)), ts);
  if (rc) return rc;
  MPN_CHECK_HIP(hipStreamSynchronize(ts));
  const int k = *n;
  if (k < 0 || k > m) { set_error("mpn_nms_host: kept count %d out of range", k); return MPN_EHIP; }
  *n_keep = k;
  memcpy(h_keep, out, sizeof(float) * 5 * (size_t)k);
  if (h_keep_idx) memcpy(h_keep_idx, idx, sizeof(int) * (size_t)k);
  return MPN_OK;
}

extern "C" int mpn_nms_host(const float *h_scored, int m, float thr, float *h_keep, int *h_keep_idx, int *n_keep) {
  MPN_CHECK_ARG(m >= 0 && n_keep != nullptr);
  *n_keep = 0;
  if (m == 0) return MPN_OK;
  MPN_CHECK_ARG(h_scored != nullptr && h_keep != nullptr);
  if (m <= kFusedMax && g_nms_fused && g_nms_force_exact == 0) return nms_host_small(h_scored, m, thr, h_keep, h_keep_idx, n_keep);
  float *d_in = nullptr, *d_keep = nullptr;
  int *d_idx = nullptr, *d_n = nullptr;
  size_t bytes = sizeof(float) * 5 * (size_t)m;
  hipStream_t ts = nullptr;
  { int rc_s = host_nms_stream(&ts); if (rc_s) return rc_s; }
  MPN_CHECK_HIP(hipMalloc(&d_in, bytes * 2 + sizeof(int) * ((size_t)m + 1)));
  d_keep = d_in + 5 * (size_t)m;
  d_idx = reinterpret_cast<int *>(d_keep + 5 * (size_t)m);
  d_n = d_idx + m;
  int rc = MPN_OK;
  hipError_t e = hipMemcpy(d_in, h_scored, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = mpn_nms(d_in, m, thr, d_keep, d_idx, d_n, ts);   // (the blocking hipMemcpy above has completed; the thread's stream does not wait on the NULL stream)
    if (rc == MPN_OK) e = hipStreamSynchronize(ts);
    if (rc == MPN_OK && e == hipSuccess) e = hipMemcpy(n_keep, d_n, sizeof(int), hipMemcpyDeviceToHost);
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

extern "C" int mpn_bbox_vote_batched(const float *d_keep, const int *d_n_keep, const float *d_scored, const int *d_counts, int n_cls,
                                     int m_stride, float thr, float score_pow, float *d_res, void *stream) {
  MPN_CHECK_ARG(n_cls >= 0 && m_stride >= 0);
  if (n_cls == 0 || m_stride == 0) return MPN_OK;
  MPN_CHECK_ARG(d_keep && d_n_keep && d_scored && d_res);
  hipLaunchKernelGGL(bbox_vote_batched_kernel, dim3(cdiv(m_stride, kWave), n_cls), dim3(kWave), 0, as_stream(stream), d_keep, d_n_keep,
                     d_scored, d_counts, m_stride, thr, score_pow, d_res);
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
