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
// Classes with up to this many tied adjacent pairs take the chunked scan with position replay (flag 3).  0 = never by default:
// measured on MI355X (tools/nms_trace.py) the replay — per-pick VALU <-> SALU round trips of ~40 cycles each — costs more than
// the slot-emulating kernel it was meant to undercut (850 k vs 320 k cycles for 1000 boxes with four tied pairs); it stays in the
// library as a second, independent exact implementation (test path 3) until its per-pick work is vectorised.
constexpr int kFewTies = 0;

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
    if (s > -10000000.0f) ++local_sel;  // nms.c:75: never picked otherwise; sorted => a prefix
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
    flags[cls] = f;
    n_sel[cls] = nsel;
  }
}

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
  const int w = blockIdx.x, rb = blockIdx.y;
  if (w * 64 >= ncol || rb * 256 >= n) return;
  if (!full && rb * 256 > w * 64 + 63) return;  // strictly lower triangle: never read by the chunked scan
  const float4 *b = sbox + (size_t)cls * m_stride;
  if (threadIdx.x < 64) {
    int j = w * 64 + threadIdx.x;
    cols[threadIdx.x] = j < ncol ? b[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int i = rb * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = b[i];
  unsigned long long bits = 0;
  const int jn = min(64, ncol - w * 64);
  for (int jj = 0; jj < jn; ++jj) {
    const int j = w * 64 + jj;
    // the chunk's own (diagonal) word is always symmetric: the scan resolves a chunk by a fixpoint over "no kept lower rank
    // overlaps me", which reads the lower triangle of that word
    if ((full || (i >> 6) == w) ? (j == i) : (j <= i)) continue;
    const float4 c = cols[jj];
    float iou = iou_plus1(a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w);  // overlap(best, other), nms.c:92
    if (!(iou <= thr)) bits |= 1ull << jj;
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

// Chunked scan, with the reference's position history replayed when the class has (a few) bit-equal scores.
//
// Tie-free class (flag 0): the pick order is the rank order, a chunk of 64 ranks is resolved with wave-uniform bit
// arithmetic on its diagonal word and the kept rows are OR-ed into the per-lane `removed` words.
//
// Class with a few tied pairs (flag 3): nms.c picks, among bit-equal scores, the box that sits FIRST in its array
// (nms.c:74-81), and the array is permuted by every round — the old first element takes the picked box's place
// (nms.c:83-85), the survivors keep their order (nms.c:91-98).  Which tied box comes first therefore depends on the whole
// history, so the history is replayed — but cheaply, and only the part that matters:
//   * positions ("slots") only ever matter when two alive boxes tie.  A chunk in which no alive rank has an alive-tied
//     successor (alive & tie word == 0) is resolved exactly as in the tie-free case;
//   * the replay of a pick needs the round's head = the alive box with the smallest slot.  A round vacates the head's slot and
//     moves boxes only to LATER slots, so the head slot strictly increases: one pointer sweeps the slots once per class.
//     "Alive at round i" is a comparison with the box's death round, recorded when a row is folded in; the sweep tests 64
//     slots per ballot from a register window.  Cost per pick: a ballot, two readlanes and (if the head is not the pick)
//     two LDS writes — instead of the slot-emulating kernel's chain of dependent LDS reads;
//   * a chunk whose first alive ranks DO tie falls back, pick by pick, to the exact rule (min slot among the alive members
//     of the run, rows applied one at a time) until the ambiguity is gone.
// LDS (flag 3 only): int16 pos[rank] / occ[slot] / death[rank].  Rows are full (symmetric) for flag 3: a tied box picked
// before a lower-ranked member of its run must still suppress it.
constexpr short kAliveForever = 0x7fff;
template <int WPL>  // 64-bit `removed` words per lane: covers m <= 4096 * WPL
__global__ __launch_bounds__(64) void nms_scan_kernel(const float4 *__restrict__ sbox, const float *__restrict__ sscore,
                                                      const int *__restrict__ sidx, const int *__restrict__ n_sel,
                                                      const int *__restrict__ flags, const int *__restrict__ counts, int m_stride, int w64,
                                                      const unsigned long long *__restrict__ mask, float *__restrict__ keep,
                                                      int *__restrict__ keep_idx, int *__restrict__ n_keep, int m_cap,
                                                      unsigned long long *__restrict__ trace) {
  extern __shared__ __attribute__((aligned(16))) short lds16[];  // pos[m_cap], occ[m_cap], death[m_cap]   (flag 3)
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
  short *pos = lds16, *occ = lds16 + m_cap, *death = lds16 + 2 * m_cap;
  unsigned long long removed[WPL], tw[WPL];
#pragma unroll
  for (int h = 0; h < WPL; ++h) { removed[h] = 0; tw[h] = 0; }
  if (replay) {
    for (int r = lane; r < m; r += kWave) {
      const int x = si[r];
      pos[r] = (short)x; occ[x] = (short)r; death[r] = kAliveForever;
    }
    for (int r0 = 0; r0 < 64 * 64 * WPL && r0 < m; r0 += kWave) {  // bit r of the tie words: ranks r and r+1 are pickable and carry the same score
      const int r = r0 + lane;
      const bool tie = (r + 1 < n) && (sc[r] == sc[r + 1]);
      const unsigned long long bal = __ballot(tie);
      if (lane == 0) tiew[r0 >> 6] = bal;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < WPL; ++h) tw[h] = (lane + 64 * h) * 64 < m ? tiew[lane + 64 * h] : 0ull;
  }
  // replay state: slots < hp are vacated or hold dead boxes; the register window caches occ / death of slots [W0, W0 + 64)
  int hp = 0, W0 = -64;
  int wf = -1, wd = 0;
  auto window_load = [&](int w0) {
    W0 = w0;
    const int sl = w0 + lane;
    wf = sl < m ? (int)occ[sl] : -1;
    wd = wf >= 0 ? (int)death[wf] : 0;
  };
  // one round of nms.c:74-85 on the slot model: `rank` was picked in round i (1-based)
  // `sb` = the pick's slot.  Returns ((f - chunk_base) << 16 | sb) when the moved box f is a rank of the current chunk (whose
  // slots the caller caches in registers), else -1.
  auto head_step = [&](int i, int rank, int sb, int chunk_base) -> int {
    int hs, f, fd;
    for (;;) {
      if (hp - W0 >= 64 || hp < W0) window_load(hp & ~63);
      const unsigned long long cand = __ballot(wf >= 0 && wd >= i) & (~0ull << (hp - W0));
      if (cand) {
        const int l = __builtin_ctzll(cand);
        hs = W0 + l;
        f = __builtin_amdgcn_readlane(wf, l);
        fd = __builtin_amdgcn_readlane(wd, l);
        break;
      }
      hp = W0 + 64;
    }
    hp = hs + 1;
    if (f == rank) return -1;
    // boxes[0] <-> boxes[best]: the old head takes the pick's slot
    if (lane == 0) { occ[sb] = (short)f; pos[f] = (short)sb; }
    if (sb - W0 < 64 && lane == sb - W0) { wf = f; wd = fd; }
    return (chunk_base >= 0 && f >= chunk_base && f < chunk_base + 64) ? (((f - chunk_base) << 16) | sb) : -1;
  };
  auto word_of = [&](int w) -> unsigned long long {  // removed word w (wave-uniform index)
    unsigned long long v = readlane64(removed[0], w & 63);
    if constexpr (WPL > 1) { if (w >= 64) v = readlane64(removed[1], w & 63); }
    return v;
  };
  auto tie_word = [&](int w) -> unsigned long long {
    unsigned long long v = readlane64(tw[0], w & 63);
    if constexpr (WPL > 1) { if (w >= 64) v = readlane64(tw[1], w & 63); }
    return v;
  };
  // fold one row into `removed`, recording the death round of every box it newly removes (replay only)
  auto record_deaths = [&](unsigned long long newly, int word, int round) {
    while (newly) {
      const int bit = __builtin_ctzll(newly);
      newly &= newly - 1;
      death[word * 64 + bit] = (short)round;
    }
  };
  int kept = 0;
  const int nchunks = (n + 63) >> 6;
  // the chunk's diagonal word, one row per lane, fetched a chunk ahead (its latency would otherwise sit in front of every chunk)
  unsigned long long diag_cur = (lane < n) ? mk[(size_t)lane * w64] : 0ull;
  for (int c = 0; c < nchunks; ++c) {
    const int base = c << 6;
    const unsigned long long diag = diag_cur;
    if (c + 1 < nchunks) diag_cur = (base + 64 + lane < n) ? mk[(size_t)(base + 64 + lane) * w64 + (c + 1)] : 0ull;
    const int nv = min(64, n - base);
    const unsigned long long valid = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
    // A chunk is resolved in passes: alive ranks below the first alive rank that carries a tie bit go through the tie-free
    // rule (bit arithmetic on the diagonal word, rows folded in a batch, heads replayed from registers); a first alive rank
    // WITH a tie bit is resolved by the exact pick-by-pick rule; repeat until the chunk is done.
    for (;;) {
      const unsigned long long rem_c = word_of(c);
      unsigned long long alive_all = ~rem_c & valid;
      if (!alive_all) break;
      const unsigned long long tiesel = replay ? (alive_all & tie_word(c)) : 0ull;
      const int first = __builtin_ctzll(alive_all);
      if (tiesel && __builtin_ctzll(tiesel) == first) {
        // ---- exact rule for one pick: among the alive members of r0's equal-score run, the one sitting first in the array
        const int r0 = base + first;
        int e = r0;  // last rank of the run = the first rank >= r0 whose tie bit is clear
        for (;;) {
          const unsigned long long ones = tie_word(e >> 6) >> (e & 63);
          const int span = 64 - (e & 63);
          const int cnt = (~ones) ? __builtin_ctzll(~ones) : 64;
          if (cnt < span) { e += cnt; break; }
          e += span;
          if (e >= n) { e = n - 1; break; }
        }
        int bp = 0x7fffffff, br = -1;
        for (int rr = r0; rr <= e; rr += kWave) {
          const int r = rr + lane;
          bool ok = r <= e;
          const int w = (ok ? r : r0) >> 6;
          unsigned long long rw = shfl64(removed[0], w & 63);
          if constexpr (WPL > 1) { const unsigned long long r1 = shfl64(removed[1], w & 63); if (w >= 64) rw = r1; }
          ok = ok && !((rw >> (r & 63)) & 1ull);
          const int p = ok ? (int)pos[r] : 0x7fffffff;
          if (p < bp) { bp = p; br = r; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const int op = __shfl_xor(bp, off), orr = __shfl_xor(br, off);
          if (op < bp) { bp = op; br = orr; }
        }
        const int pick = __builtin_amdgcn_readfirstlane(br);
        const int round = kept + 1;
        if (lane == 0) {
          const float4 bx = b[pick];
          float *q = kout + (size_t)kept * 5;
          q[0] = bx.x; q[1] = bx.y; q[2] = bx.z; q[3] = bx.w; q[4] = sc[pick];
          if (kidx) kidx[kept] = si[pick];
          death[pick] = (short)round;
        }
#pragma unroll
        for (int h = 0; h < WPL; ++h) {
          const int w = lane + 64 * h;
          if (w < w64 && w * 64 < m) {
            const unsigned long long row = mk[(size_t)pick * w64 + w];
            unsigned long long newly = row & ~removed[h];
            if (w == (pick >> 6)) newly &= ~(1ull << (pick & 63));
            record_deaths(newly, w, round);
            removed[h] |= row;
            if (w == (pick >> 6)) removed[h] |= 1ull << (pick & 63);
          }
        }
        __syncthreads();
        window_load(hp & ~63);  // deaths changed: refresh the cached window
        head_step(round, pick, __builtin_amdgcn_readfirstlane((int)pos[pick]), -1);
        __syncthreads();
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
      if (replay) {
        // death rounds of this chunk's ranks, all lanes at once: a picked rank dies in its own round; a suppressed one in the round of
        // the FIRST kept rank that overlaps it — the rows are symmetric (flag 3), so that is the lowest set bit of
        // (own row & kept ranks below it)
        const bool was_alive = (alive_all >> lane) & 1ull;
        const unsigned long long below = (1ull << lane) - 1ull;
        const bool picked = (keptmask >> lane) & 1ull;
        const unsigned long long killers = diag & keptmask & below;
        if (was_alive && (picked || killers)) {
          const int kr = picked ? lane : __builtin_ctzll(killers);
          death[base + lane] = (short)(kept + __popcll(keptmask & ((1ull << kr) - 1ull)) + 1);
        }
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
      }
      // fold the kept rows into `removed`: words after this chunk, and the chunk's own word (resolved range + in-chunk kills)
      NMS_STAMP(3);
      if (replay) {
        unsigned long long run = removed[0];
        int rnd = kept;
#pragma unroll
        for (int r = 0; r < 64; ++r) {
          if ((keptmask >> r) & 1ull) {  // wave-uniform
            ++rnd;
            record_deaths(rows[r] & ~run, lane, rnd);
            run |= rows[r];
          }
        }
        removed[0] = run;
      } else {
        unsigned long long acc = 0ull;
#pragma unroll
        for (int r = 0; r < 64; ++r) acc |= ((keptmask >> r) & 1ull) ? rows[r] : 0ull;
        removed[0] |= acc;
      }
      if (lane == c) removed[0] |= (valid & lim_mask) | (diag_acc & valid);
      if constexpr (WPL > 1) {  // second word per lane (m > 4096): non-speculative, batched
        const int wsec = lane + 64;
        if (wsec < w64 && wsec > c) {
          unsigned long long km = keptmask;
          int rnd = kept;
          while (km) {
            const int r = __builtin_ctzll(km);
            km &= km - 1;
            ++rnd;
            const unsigned long long row = mk[(size_t)(base + r) * w64 + wsec];
            if (replay) record_deaths(row & ~removed[1], wsec, rnd);
            removed[1] |= row;
          }
        }
        if (wsec == c) removed[1] |= (valid & lim_mask) | (diag_acc & valid);
      }
      NMS_STAMP(4);
      if (replay) {
        __syncthreads();
        window_load(hp & ~63);
        int posr = base + lane < m ? (int)pos[base + lane] : 0;  // slots of this chunk's ranks, kept current in registers
        // Heads of this chunk's rounds.  Eligibility masks (slot holds a box alive at the round) for 8 rounds at a time are 8
        // independent ballots; the sweep itself is SALU; the moved box and the pick's slot are read with readlane only to feed LDS
        // writes and a rare-path test (a move INTO the register window, or the window running out).
        unsigned long long km = keptmask;
        int rnd = kept;
        while (km) {
          unsigned long long elig[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) elig[q] = __ballot(wf >= 0 && wd >= rnd + 1 + q);
          bool redo = false;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (km && !redo) {
              const int sh = hp - W0;
              const unsigned long long cand = sh >= 64 ? 0ull : (elig[q] & (~0ull << sh));
              if (!cand) { redo = true; hp = W0 + 64; window_load(hp & ~63); }   // window exhausted: reload and recompute the masks
              else {
                const int r = __builtin_ctzll(km);
                km &= km - 1;
                ++rnd;
                const int l = __builtin_ctzll(cand);
                const int f = __builtin_amdgcn_readlane(wf, l), fd = __builtin_amdgcn_readlane(wd, l);
                const int sb = __builtin_amdgcn_readlane(posr, r);
                hp = W0 + l + 1;
                // boxes[0] <-> boxes[best] (nms.c:83-85): the old head takes the pick's slot (a no-op when the head IS the pick)
                if (lane == 0) { occ[sb] = (short)f; pos[f] = (short)sb; }
                if (lane == f - base) posr = sb;
                if (sb - W0 < 64) {  // rare: the move lands inside the register window -> later rounds' masks change
                  if (lane == sb - W0) { wf = f; wd = fd; }
                  redo = true;
                }
              }
            }
          }
        }
        __syncthreads();
      }
      NMS_STAMP(5);
      kept += __popcll(keptmask);
      if (limit == 64) break;
    }
  }
  if (lane == 0) n_keep[cls] = kept;
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

MPN_KNOB(int, g_nms_force_exact, 0);  // test hook: 1 = always the exact IoU-sweep kernel, 2 = always the tie (slot-emulation) kernel, 3 = always the replaying scan
MPN_KNOB(unsigned long long *, g_nms_trace, nullptr);
#ifdef MPN_DEBUG_HOOKS
extern "C" void mpn_debug_set_nms_force_exact(int v) { g_nms_force_exact = v; }
extern "C" void mpn_debug_set_nms_trace(void *p) { g_nms_trace = static_cast<unsigned long long *>(p); }
#endif

extern "C" int mpn_nms_batched(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr,
                               float *d_keep, int *d_keep_idx, int *d_n_keep, void *stream) {
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
    hipLaunchKernelGGL(nms_mask_kernel, dim3(w64, cdiv(m_stride, 256), n_cls), dim3(256), 0, st, sbox, n_sel, flags, d_counts,
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
      const size_t slds = (size_t)3 * scap * sizeof(short);
      int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<1>), 3 * 4096 * 2);
      if (rc_attr == MPN_OK) rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(nms_scan_kernel<2>), 3 * 8192 * 2);
      if (rc_attr) return rc_attr;
      if (w64 <= 64)
        hipLaunchKernelGGL(nms_scan_kernel<1>, dim3(n_cls), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, d_counts, m_stride, w64, mask,
                           d_keep, d_keep_idx, d_n_keep, scap, g_nms_trace);
      else
        hipLaunchKernelGGL(nms_scan_kernel<2>, dim3(n_cls), dim3(kWave), slds, st, sbox, sscore, sidx, n_sel, flags, d_counts, m_stride, w64, mask,
                           d_keep, d_keep_idx, d_n_keep, scap, g_nms_trace);
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
