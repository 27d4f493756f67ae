// dense.hip — fp32-MFMA dense kernels for gfx950: 3x3 convolution (implicit GEMM, fused bias + ReLU +
// optional ceil-mode 2x2 max-pool), linear (GEMM, fused bias + ReLU, deterministic split-K), ROI
// max-pool into the GEMM operand layout, max-pool, layout converters and weight packers.
//
// Why fp32 MFMA: north_star fixes fp32 results (1e-4 on scores); gfx950 has no TF32/xf32, and
// v_mfma_f32_32x32x2_f32 is exact fp32 at the full 157 TFLOP/s vector rate (an fmaf chain).
//
// Kernel shape (both conv and GEMM):
//   * 256 threads = 4 waves, each wave owns a 64 x 64 output tile = 2x2 MFMA 32x32 accumulators
//     (64 VGPRs), M side = output channels / weight rows, N side = pixels / matrix rows;
//   * operands are staged HBM -> LDS as LINEAR copies (global_load_lds, 16 B per lane) of the
//     channel-blocked HBM layouts described in dense.h, double-buffered: the next K chunk's DMA is
//     issued before the current chunk's 144 (conv) / 64 (GEMM) MFMAs per wave and has thousands of
//     cycles to land, so one barrier per chunk is the only synchronisation;
//   * every MFMA operand fetch is one ds_read_b128 whose 64 lanes cover 1 KiB contiguous LDS
//     (conflict-free) and feeds 4 MFMAs;
//   * D[cout][pixel] leaves the accumulators as float4 stores that are 1 KiB-contiguous per wave
//     instruction in the C8P layout — the next layer's LDS image.
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dense.h"

namespace mpn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MPN_GPTR(p) ((const __attribute__((address_space(1))) void *)(p))
#define MPN_LPTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  // 64 lanes x 16 B -> lds_wave_base[lane*4 .. lane*4+3]; lds_wave_base must be wave-uniform.
  __builtin_amdgcn_global_load_lds(MPN_GPTR(gsrc), MPN_LPTR(lds_wave_base), 16, 0, 0);
}

// LDS-DMA with the address split the way the hardware takes it: wave-uniform 64-bit base in SGPRs + one 32-bit per-lane
// byte offset (the SADDR form).  The builtin always materialises a 64-bit per-lane address with VALU instructions, and on
// this chip VALU work from a wave does NOT overlap its MFMAs (tools/probes/mfma_shadow.cpp: every VALU instruction
// between two MFMAs costs its issue time plus ~20 cycles for breaking the MFMA stream), so the inner loops issue their
// DMA through this.  The compiler does not know these loads exist: callers wait with dma_wait_all() before the barrier
// that publishes the data (and get counted lgkmcnt waits instead of the lgkmcnt(0) it forces after a builtin LDS-DMA).
__device__ __forceinline__ void glds16_saddr(const float *base_uniform, unsigned lane_byte_off, unsigned lds_byte_addr_uniform) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(lds_byte_addr_uniform), "v"(lane_byte_off), "s"(base_uniform)
               : "memory");  // M0 is not in the clobber list (reserved): kernels that use this issue ALL their LDS-DMA through it
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_byte_addr(const float *p) {
  return (unsigned)(size_t)((__attribute__((address_space(3))) const float *)p);
}

// =================================================================================================
// conv3x3, stride 1, pad 1, C8P in/out
// =================================================================================================
struct ConvArgs {
  const float *in; size_t in_plane; int in_Wp;
  const float *wpk; int CoutP; const float *bpk;
  float *out; size_t out_plane; int out_Wp;
  float *pool; size_t pool_plane; int pool_Wp; int pool_H, pool_W;
  int H, W, nchunks, out_cb, relu, n_ct, tiles_x;
  // split-K (blockIdx.y = split): raw partial sums go to part + split*part_slab in the OUTPUT's C8P
  // geometry; conv_splitk_reduce_kernel adds them in split order and applies bias/ReLU/pool.
  int splits, chunks_per_split;
  float *part; size_t part_slab;
  // Winograd kernel, tail split: blocks [0, tail_first) run their whole K range and finish their tiles themselves; every
  // tile from tail_first on (the launch's last, partly filled round of CUs) is cut into tail_splits K ranges of tail_cps
  // chunks that write partial slabs — the tail round then takes ~1/tail_splits of a full block's time.  0 = off.
  int tail_first, tail_splits, tail_cps;
  int wino_tc;  // Winograd block geometry: 8 = 8 x 8 tiles (16 x 16 px), 16 = 4 x 16 tiles (8 x 32 px)
  int ablate;  // timing experiments only (results wrong): 1 = no DMA in loop, 2 = no barrier, 4 = no ds_reads
  unsigned long long *trace;  // tools/wino_trace.py: per-chunk s_memtime stamps of wave 0 of blocks 0..3 (Winograd kernel, ABL 64)
};

// TPS = taps per LDS stage: 9 -> one stage per 8-channel chunk (input tile + all 9 taps' weights);
//                           3 -> three stages per chunk (one filter row each): the weight ring shrinks to
//                                2 x 3 taps, LDS drops from 88 KB to ~39 KB per block and 3 blocks share a CU,
//                                so one block's prologue / epilogue / barrier gaps hide behind the others' MFMAs.
template <int BM, int TH, int WM, int WN, int TPS>
__global__ __launch_bounds__(256) void conv3x3_c8p_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TPS == 9 || TPS == 3, "taps per stage");
  constexpr int MI = BM / WM / 32, NI = TH / WN;
  static_assert(MI >= 1 && NI >= 1, "tile");
  constexpr int G = 9 / TPS;                        // stages per chunk
  constexpr int IN_PIECES = (TH + 2) * 68;          // 16-byte pieces of the (TH+2) x 34 px halo tile
  constexpr int IN_LOADS = (IN_PIECES + 63) / 64;   // 1 KiB wave-loads
  constexpr int IN_FLOATS = IN_LOADS * 256;
  constexpr int W_LOADS = TPS * BM / 32;
  constexpr int W_FLOATS = TPS * BM * 8;
  constexpr int IN_IT = (IN_LOADS + 3) / 4, W_IT = (W_LOADS + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [IN x2][W x2]
  float *const in_lds = lds;
  float *const w_lds = lds + 2 * IN_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ct = blockIdx.x % a.n_ct, sp = blockIdx.x / a.n_ct;
  const int ty = sp / a.tiles_x, tx = sp - ty * a.tiles_x;
  const int y0 = ty * TH, x0 = tx * 32, cout0 = ct * BM;
  const int wm = wave / WN, wn = wave % WN;
  const int mbase = wm * (BM / WM), rbase = wn * NI;

  // ---- per-thread staging geometry (constant over the K loop)
  int in_off[IN_IT]; bool in_ok[IN_IT];
#pragma unroll
  for (int i = 0; i < IN_IT; ++i) {
    int t = i * 4 + wave, p = t * 64 + lane;
    in_ok[i] = (t < IN_LOADS) && (p < IN_PIECES);
    int r = p / 68, o = p - r * 68;
    in_off[i] = ((y0 + r) * a.in_Wp + x0) * 8 + o * 4;
  }
  int w_off[W_IT]; bool w_ok[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    int t = i * 4 + wave, p = t * 64 + lane;
    w_ok[i] = t < W_LOADS;
    int tap = p / (BM * 2), rem = p - tap * (BM * 2);  // tap local to the stage
    w_off[i] = (tap * a.CoutP + cout0) * 8 + rem * 4;
  }
  const size_t w_stage = (size_t)TPS * a.CoutP * 8;  // packed weights advance by TPS taps per stage

  // one staging item = one 1 KiB wave-load: weight items first, then (first stage of a chunk only) input items
  auto issue_item = [&](int q, int item) {
    if (item < W_IT) {
      const int i = item;
      if (w_ok[i]) glds16(a.wpk + (size_t)q * w_stage + w_off[i], w_lds + (q & 1) * W_FLOATS + (i * 4 + wave) * 256);
    } else if (item - W_IT < IN_IT) {
      const int i = item - W_IT;
      const int c = q / G;
      if ((q - c * G) == 0 && in_ok[i]) glds16(a.in + (size_t)c * a.in_plane + in_off[i], in_lds + (c & 1) * IN_FLOATS + (i * 4 + wave) * 256);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int lane_off = l31 * 8 + half * 4;
  const int c0 = blockIdx.y * a.chunks_per_split;
  const int c1 = min(a.nchunks, c0 + a.chunks_per_split);
  const int q0 = c0 * G, q1 = c1 * G;  // stage index q = chunk * G + filter-row group
  constexpr int ITEMS = W_IT + IN_IT;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) issue_item(q0, it);
  __syncthreads();

  constexpr int PER_TAP = (ITEMS + TPS - 1) / TPS;  // DMA items issued per tap: the next stage is fully in flight by the last tap
  for (int q = q0; q < q1; ++q) {
    const int c = q / G, g = q - c * G;
    const bool more = (q + 1 < q1) && !MPN_ABLATE(a.ablate & 1);
    const float *Il = in_lds + (c & 1) * IN_FLOATS + lane_off + (TPS == 3 ? g * 34 * 8 : 0);
    const float *Wl = w_lds + (q & 1) * W_FLOATS + lane_off;
    // operand fragments are double-buffered in registers: tap t+1's ds_reads are issued before tap t's
    // 16 MFMAs, so LDS latency never sits on the matrix pipe; the next stage's DMA is spread over the taps.
    f32x4 af[2][MI], bf[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[0][mi] = *reinterpret_cast<const f32x4 *>(Wl + (mbase + mi * 32) * 8);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[0][ni] = *reinterpret_cast<const f32x4 *>(Il + ((rbase + ni) * 34) * 8);
    if (MPN_ABLATE(a.ablate & 4)) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[1][mi] = af[0][mi];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[1][ni] = bf[0][ni];
    }
#pragma unroll
    for (int t = 0; t < TPS; ++t) {
      const int cur = t & 1;
      if (t + 1 < TPS && !MPN_ABLATE(a.ablate & 4)) {
        const int nt = t + 1, dy = (TPS == 9) ? nt / 3 : 0, dx = nt % 3;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[cur ^ 1][mi] = *reinterpret_cast<const f32x4 *>(Wl + (nt * BM + mbase + mi * 32) * 8);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bf[cur ^ 1][ni] = *reinterpret_cast<const f32x4 *>(Il + ((rbase + ni + dy) * 34 + dx) * 8);
      }
      if (more) {
#pragma unroll
        for (int k = 0; k < PER_TAP; ++k) issue_item(q + 1, t * PER_TAP + k);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mi][j], bf[cur][ni][j], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!MPN_ABLATE(a.ablate & 2)) __syncthreads();
  }

  const int x = x0 + l31;
  const bool xok = x < a.W;
  if (a.splits > 1) {  // raw partial sums; the reduce kernel finishes the layer
    float *pb = a.part + (size_t)blockIdx.y * a.part_slab;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = (cout0 + mbase + mi * 32) / 8 + g;
        if (cb >= a.out_cb) continue;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int y = y0 + rbase + ni;
          if (xok && y < a.H) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][g * 4 + e];
            *reinterpret_cast<f32x4 *>(pb + (size_t)cb * a.out_plane + ((size_t)(y + 1) * a.out_Wp + x + 1) * 8 + half * 4) = v;
          }
        }
      }
    return;
  }
  // ---- epilogue: bias + ReLU, C8P float4 stores, optional fused ceil-mode 2x2 max-pool
  if (MPN_ABLATE(a.ablate & 8)) {  // timing experiment: skip the output stores (the never-true store keeps the accumulators live)
    if (a.H < 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) *reinterpret_cast<f32x16 *>(a.part + (size_t)(mi * NI + ni) * 16 + lane * 64) = acc[mi][ni];
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cb = (cout0 + mbase + mi * 32) / 8 + g;
      if (cb >= a.out_cb) continue;
      const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4);
      f32x4 v[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int y = y0 + rbase + ni;
        const bool ok = xok && (y < a.H);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[mi][ni][g * 4 + e] + b4[e];
          if (a.relu) t = t < 0.0f ? 0.0f : t;
          v[ni][e] = t;
        }
        if (ok && a.out)
          *reinterpret_cast<f32x4 *>(a.out + (size_t)cb * a.out_plane + ((size_t)(y + 1) * a.out_Wp + x + 1) * 8 + half * 4) = v[ni];
        if (!ok) v[ni] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
      if constexpr (NI % 2 == 0) {
        if (a.pool) {  // rows (y0+rbase+2q, +1) are a vertical pooling pair; lanes (2j,2j+1) a horizontal one
#pragma unroll
          for (int q = 0; q < NI / 2; ++q) {
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = fmaxf(v[2 * q][e], v[2 * q + 1][e]);
              m[e] = fmaxf(t, __shfl_xor(t, 1));
            }
            const int py = ((y0 + rbase) >> 1) + q, px = x >> 1;
            if (!(l31 & 1) && py < a.pool_H && px < a.pool_W)
              *reinterpret_cast<f32x4 *>(a.pool + (size_t)cb * a.pool_plane + ((size_t)(py + 1) * a.pool_Wp + px + 1) * 8 + half * 4) = m;
          }
        }
      }
    }
  }
}

MPN_KNOB(int, g_gemm_ablate, 0);         // timing-experiment switch of the convolution kernels (tools/ablate_conv.py, tools/ablate_wino.py)

// =================================================================================================
// First layer (<= 4 input channels; VGG conv1_1: 3 -> 64 on the full-resolution image).
// The generic kernel spends 9 taps x 8-channel chunks = 72 K-steps on a layer whose real K is 27, and the layer is
// HBM-bound on its output (154 MB at 600 x 1000): here K = 9 taps x 4 channels = 36 (18 MFMA k-pairs, the 4th channel is
// the C8P record's zero pad), the 36 x 64 weight block and the bias live in LDS for the whole (persistent) block, the
// (TH+2) x 34 x 4-channel input tile is staged into LDS as channel planes (conflict-free ds_read_b32: lanes = consecutive
// pixels), double-buffered, and 20.5 KB of LDS + 71 VGPRs let five blocks share a CU so that one block's stores overlap
// another's MFMAs (see the kernel body for the ordering rules that make that overlap actually happen).
// Packed weights: w36[(tap*2 + p)*2 + half][CoutP] = w[cout][cin = 2p + half][tap]  (cin >= Cin -> 0).
// =================================================================================================
constexpr int kF_TH = 8;                       // rows per block tile (4 waves x 2 rows), 32 columns
constexpr int kF_PL = (kF_TH + 2) * 34 + 12;   // floats per channel plane of the LDS tile (352: 16-byte multiple)
template <int ABL>  // ABL: compile-time timing-experiment switches (0 in production; tools/ablate_first.py): 1 no stores, 2 no MFMAs
__global__ __launch_bounds__(256, 5) void conv3x3_first_kernel(const float *__restrict__ in, int in_Wp, const float *__restrict__ w36,
                                                               int CoutP, const float *__restrict__ bpk, float *__restrict__ out,
                                                               size_t out_plane, int out_Wp, int H, int W, int out_cb, int relu,
                                                               int tiles_x, int n_tiles) {
  // Persistent over tiles (grid = 4 blocks per CU): the 36 x 64 weight block is staged ONCE into LDS, and the input tile of the
  // block's next tile is fetched into registers before the current tile's MFMAs and parked in the other LDS buffer after them —
  // the layer is latency-bound otherwise (one block = weight loads -> tile load -> 72 MFMAs -> stores, nothing overlapping
  // within the block).  Operands come from LDS per tap (A: one ds_read2_b32, B: two ds_read_b32 per k-pair), which keeps the
  // kernel under 128 VGPRs = 4 waves per SIMD.
  __shared__ float tile[2][4 * kF_PL];
  __shared__ float wl[36 * 64];
  __shared__ float bl[64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int cout0 = blockIdx.y * 64;
  for (int i = tid; i < 36 * 64; i += 256) wl[i] = w36[(size_t)(i >> 6) * CoutP + cout0 + (i & 63)];
  if (tid < 64) bl[tid] = bpk[cout0 + tid];
  constexpr int NPX = (kF_TH + 2) * 34;  // 340 pixel records per tile: at most 2 per thread
  const int p0 = tid, p1 = tid + 256;
  const int r0 = p0 / 34, c0 = p0 - r0 * 34, r1 = p1 / 34, c1 = p1 - r1 * 34;
  const int off0 = (r0 * in_Wp + c0) * 8, off1 = p1 < NPX ? (r1 * in_Wp + c1) * 8 : off0;
  auto fetch = [&](int t, f32x4 &v0, f32x4 &v1) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const float *base = in + ((size_t)(ty * kF_TH) * in_Wp + tx * 32) * 8;
    v0 = *reinterpret_cast<const f32x4 *>(base + off0);
    v1 = *reinterpret_cast<const f32x4 *>(base + off1);
  };
  auto park = [&](int buf, const f32x4 &v0, const f32x4 &v1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[buf][e * kF_PL + p0] = v0[e];
    if (p1 < NPX) {
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[buf][e * kF_PL + p1] = v1[e];
    }
  };
  // vmcnt counts loads and stores in issue order on this ISA: a wait for a load also waits for every OLDER store.  So nothing in
  // the loop waits on a global load younger than the tile's stores (bias comes from LDS; the next input tile is fetched BEFORE
  // this tile's stores and parked a whole MFMA phase later), and a tile's 16 stores drain under the next tile's MFMAs.
  int t = blockIdx.x;
  if (t >= n_tiles) return;
  const int step = gridDim.x;
  f32x4 n0, n1;
  fetch(t, n0, n1);
  park(0, n0, n1);
  if (t + step < n_tiles) fetch(t + step, n0, n1);
  __syncthreads();
  const float *wa = wl + half * 64 + l31;
  int buf = 0;
  for (; t < n_tiles; t += step, buf ^= 1) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * kF_TH, x0 = tx * 32;
    const float *tb = tile[buf] + half * kF_PL + (wave * 2) * 34 + l31;
    // output addressing: a uniform (SGPR) plane base + one 32-bit per-lane byte offset per output row (hoisted 64-bit per-store
    // pointers would cost 32 VGPRs)
    const int x = x0 + l31;
    bool xok = x < W;
    if constexpr ((ABL & 1) != 0) xok = xok && in_Wp == 7;  // never true: keeps the accumulators live
    const int yw = y0 + wave * 2;
    const unsigned voff = (unsigned)(((yw + 1) * out_Wp + x + 1) * 8 + half * 4) * 4u;  // bytes inside a channel-block plane
    const unsigned vrow = (unsigned)out_Wp * 32u;
    const bool ok0 = xok && yw < H, ok1 = xok && yw + 1 < H;
    // Two halves of 32 output channels: 36 MFMAs (two independent accumulator chains), then that half's 8 stores — the first
    // stores leave after a quarter of the block's MFMA work and the other half's MFMAs (and the other waves') run while they
    // drain; a whole-tile epilogue leaves HBM idle for the first MFMA phase and the MFMA pipe idle for the last store phase.
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      f32x16 acc[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.0f;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float afr = wa[(tp * 2 + p) * 128 + mi * 32];
          float bfr[2];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bfr[ni] = tb[2 * p * kF_PL + (ni + dy) * 34 + dx];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr ((ABL & 2) == 0) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr, bfr[ni], acc[ni], 0, 0, 0);
            else acc[ni][(tp + p) & 15] += afr * bfr[ni];
          }
        }
        if ((tp % 3) == 2) __builtin_amdgcn_sched_barrier(0);  // operand loads at most three taps ahead (VGPR budget)
      }
      if (mi == 1) {
        // the next tile: park what was fetched one tile ago (the other buffer's last readers finished before the previous
        // barrier), then start the fetch of the one after it — ahead of this half's stores
        const int tn = t + step;
        if (tn < n_tiles) {
          park(buf ^ 1, n0, n1);
          if (tn + step < n_tiles) fetch(tn + step, n0, n1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // bias + ReLU, 1 KiB-contiguous float4 stores in the next layer's C8P layout
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cb = __builtin_amdgcn_readfirstlane((cout0 + mi * 32) / 8 + g);
        if (cb >= out_cb) continue;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bl + (mi * 4 + g) * 8 + half * 4);
        char *pb = reinterpret_cast<char *>(out + (size_t)cb * out_plane);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float tv = acc[ni][g * 4 + e] + b4[e];
            if (relu) tv = tv < 0.0f ? 0.0f : tv;
            v[e] = tv;
          }
          if (ni ? ok1 : ok0) *reinterpret_cast<f32x4 *>(pb + (voff + (ni ? vrow : 0u))) = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
}

__global__ void pack_conv_w_first_kernel(const float *__restrict__ w, int Cin, int Cout, int CoutP, float *__restrict__ w36) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 36 * CoutP) return;
  const int row = t / CoutP, co = t - row * CoutP;   // row = (tap*2 + p)*2 + half
  const int hf = row & 1, ks = row >> 1, p = ks & 1, tap = ks >> 1;
  const int cin = 2 * p + hf;
  w36[t] = (co < Cout && cin < Cin) ? w[((size_t)co * Cin + cin) * 9 + tap] : 0.0f;
}

size_t conv_first_elems(int Cout) { return (size_t)36 * conv_coutp(Cout); }

int pack_conv_weights_first(const float *d_w, int Cin, int Cout, float *d_w36, hipStream_t s) {
  MPN_CHECK_ARG(d_w && d_w36 && Cin > 0 && Cin <= 4 && Cout > 0);
  const int CoutP = conv_coutp(Cout);
  hipLaunchKernelGGL(pack_conv_w_first_kernel, dim3(cdiv(36 * CoutP, 256)), dim3(256), 0, s, d_w, Cin, Cout, CoutP, d_w36);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

int conv3x3_first_c8p(Act in, const float *d_w36, const float *d_bpk, int Cout, int relu, Act out, hipStream_t s) {
  MPN_CHECK_ARG(in.p && d_w36 && d_bpk && out.p && in.C <= 4 && out.H == in.H && out.W == in.W && out.C == Cout);
  const int tiles_x = cdiv(in.W, 32), tiles_y = cdiv(in.H, kF_TH), n_ct = cdiv(Cout, 64);
  const int n_tiles = tiles_x * tiles_y;
  // persistent grid: at most 5 resident blocks per CU (71 VGPRs, 20.5 KB LDS each), every block the same number of tiles
  // (2400 tiles at 600 x 1000 -> 1200 blocks x 2 tiles)
  const char *e_cap = MPN_ABLATE(1) ? getenv("MPN_FIRST_BLOCKS") : nullptr;  // timing experiment (debug flavour only)
  const int cap = e_cap ? atoi(e_cap) : 1280;
  const int blocks = cdiv(n_tiles, cdiv(n_tiles, cap));
  auto kern = conv3x3_first_kernel<0>;
#ifdef MPN_DEBUG_HOOKS
  if ((g_gemm_ablate & 3) == 1) kern = conv3x3_first_kernel<1>;
  if ((g_gemm_ablate & 3) == 2) kern = conv3x3_first_kernel<2>;
  if ((g_gemm_ablate & 3) == 3) kern = conv3x3_first_kernel<3>;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)n_ct), dim3(256), 0, s, in.p, in.Wp, d_w36, conv_coutp(Cout), d_bpk, out.p,
                     out.plane(), out.Wp, in.H, in.W, (Cout + 7) / 8, relu, tiles_x, n_tiles);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

template <int BM, int TH, int WM, int WN, int TPS>
static int launch_conv(const ConvArgs &a0, int tiles_y, hipStream_t s) {
  ConvArgs a = a0;
  constexpr int IN_LOADS = ((TH + 2) * 68 + 63) / 64;
  constexpr size_t LDS = (size_t)2 * (IN_LOADS * 256 + TPS * BM * 8) * sizeof(float);
  auto kern = conv3x3_c8p_kernel<BM, TH, WM, WN, TPS>;
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(kern), (int)LDS); if (rc_attr) return rc_attr; }
  dim3 grid((unsigned)(a.n_ct * tiles_y * a.tiles_x), (unsigned)a.splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, a);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}


// =================================================================================================
// conv3x3 via Winograd F(2x2,3x3), fused: input transform, 16 component GEMMs on MFMA, output transform
// =================================================================================================
// Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A  per 2x2 output tile (Lavin & Gray 2015) — the algorithm cuDNN's
// WINOGRAD conv algo runs for the reference's cudnn.SpatialConvolution; 16 multiplies per 4 outputs instead of 36,
// i.e. 2.25x fewer MFMAs than the direct kernel, still all-fp32 (error ~1e-6 of the output scale).
//   * block = 64 couts x (16 x 16 output px = 8 x 8 Winograd tiles); wave = 32 couts x 32 tiles x 16 components
//     = 16 MFMA accumulators (256 AGPRs, one wave per SIMD);
//   * per 8-channel chunk: the raw 18x18 halo tile and the 16 pre-transformed weight slices [comp][cout][8] are
//     DMA'd (global_load_lds) one chunk ahead; the input transform B^T d B of chunk c+1 (thread = one tile x one
//     channel pair: 16 ds_read_b64, 32 packed adds, 16 ds_write_b64) is interleaved between chunk c's MFMA groups
//     and lands in the other V buffer [comp][tile][8];
//   * the 16 components of one (cout, tile) sit at the same (lane, register) of the 16 accumulators, so the
//     output transform A^T M A, bias, ReLU and the 2x2 max-pool (a Winograd tile IS a pooling window) are
//     register-local; float4 stores go straight into the next layer's C8P layout.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {  // a - b as ONE v_pk_add_f32 (negated second operand)
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Two block geometries over the same 64 Winograd tiles: TC = 8 -> 8 x 8 tiles = 16 x 16 output px (raw halo tile 18 x 18 px),
// TC = 16 -> 4 x 16 tiles = 8 rows x 32 columns (raw tile 10 x 34 px).  The host picks per layer whichever pads the map
// less (VGG conv5 at 38 x 63: 48 x 64 px of tiles vs 40 x 64).  Both need 11 wave-loads for the raw tile.
constexpr int WG_RAW_LOADS = 12;                      // 10.1 / 10.6 wave-loads of data; padded to 3 per wave so the DMA issue is branch-free
constexpr int WG_RAW_FLOATS = WG_RAW_LOADS * 256;
constexpr int WG_V_FLOATS = 16 * 64 * 8;
constexpr int WG_U_FLOATS = 16 * 64 * 8;
constexpr size_t WG_LDS_BYTES = (size_t)2 * (WG_RAW_FLOATS + WG_V_FLOATS + WG_U_FLOATS) * sizeof(float);

template <int ABL, int TC>  // ABL: compile-time timing-experiment switches (0 in production; see tools/ablate_wino.py); TC: tiles per block row
__global__ __launch_bounds__(256) void conv3x3_wino_kernel(ConvArgs a) {
  constexpr int TR = 64 / TC;                         // tile rows per block
  constexpr int RC = 2 * TC + 2, RR = 2 * TR + 2;     // raw halo tile: RR rows x RC px
  constexpr int WG_RAW_PIECES = RR * RC * 2;          // two 16-byte pieces per pixel record
  static_assert((WG_RAW_PIECES + 63) / 64 <= WG_RAW_LOADS, "raw tile DMA slots");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *const raw_lds = lds;
  float *const v_lds = lds + 2 * WG_RAW_FLOATS;
  float *const u_lds = v_lds + 2 * WG_V_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0;
  if constexpr ((ABL & 64) != 0) tk0 = __builtin_amdgcn_s_memtime();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x, split = blockIdx.y, cps = a.chunks_per_split;
  bool partial = a.splits > 1;
  if (a.tail_splits > 1 && bid >= a.tail_first) {
    const int t = bid - a.tail_first;
    split = t % a.tail_splits; bid = a.tail_first + t / a.tail_splits; cps = a.tail_cps; partial = true;
  }
  const int ct = bid % a.n_ct, sp = bid / a.n_ct;
  const int tyb = sp / a.tiles_x, txb = sp - tyb * a.tiles_x;
  const int y0 = tyb * (2 * TR), x0 = txb * (2 * TC), cout0 = ct * 64;
  const int mbase = (wave >> 1) * 32, nbase = (wave & 1) * 32;

  // DMA sources are (wave-uniform 64-bit base) + (32-bit per-lane byte offset): no 64-bit VALU address arithmetic in
  // the loop and one offset VGPR per item instead of a 64-bit pair
  constexpr int RAW_IT = (WG_RAW_LOADS + 3) / 4;
  unsigned raw_rel[RAW_IT];
#pragma unroll
  for (int i = 0; i < RAW_IT; ++i) {
    int p = min((i * 4 + wave) * 64 + lane, WG_RAW_PIECES - 1);  // slots past the tile re-load its last piece (never read)
    int r = p / (RC * 2), o = p - r * (RC * 2);
    raw_rel[i] = (unsigned)(((r * a.in_Wp) * 8 + o * 4) * 4);
  }
  // weight slice item i of this wave: component (i*4+wave)/2 (uniform), 16-byte piece (wave&1)*64 + lane of its 64 couts
  const unsigned u_lane = (unsigned)((((wave & 1) * 64 + lane) * 4) * 4);
  const size_t u_chunk = (size_t)16 * a.CoutP * 8;
  const float *const in_tile = a.in + (size_t)(y0 * a.in_Wp + x0) * 8;
  const float *const w_tile = a.wpk + (size_t)cout0 * 8;
  const unsigned lds0 = lds_byte_addr(lds);  // LDS byte address of the dynamic segment (wave-uniform)
  const unsigned raw_slot = lds0 + (unsigned)(wave * 256) * 4, u_slot = lds0 + (unsigned)(2 * WG_RAW_FLOATS + 2 * WG_V_FLOATS + wave * 256) * 4;
  auto issue_raw_item = [&](int c, int buf, int i) {
    glds16_saddr(in_tile + (size_t)c * a.in_plane, raw_rel[i], raw_slot + (unsigned)(buf * WG_RAW_FLOATS + i * 1024) * 4);
  };
  auto issue_raw = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < RAW_IT; ++i) issue_raw_item(c, buf, i);
  };
  auto issue_u = [&](int c, int buf, int i) {
    const float *slice = w_tile + (size_t)c * u_chunk + (size_t)(((i * 4 + wave) >> 1) * a.CoutP) * 8;  // wave-uniform
    glds16_saddr(slice, u_lane, u_slot + (unsigned)(buf * WG_U_FLOATS + i * 1024) * 4);
  };

  // input transform: wave w owns tiles 16w..16w+15 (tile rows 2w, 2w+1); lane = tile_local * 4 + channel pair, so the
  // 64 ds_write_b64 of one component cover 512 contiguous bytes and each half-wave's ds_read_b64 hits 32 distinct
  // bank pairs (a tile-per-lane mapping was 8-way bank-conflicted and made the LDS pipe the bottleneck)
  const int tf_tile = wave * 16 + (lane >> 2), tf_cp = lane & 3;
  const int tf_rd = ((2 * (tf_tile / TC)) * RC + 2 * (tf_tile % TC)) * 8 + 2 * tf_cp;
  // V / U records are [row][8 floats] = two 16-byte halves (k 0-3, k 4-7); a fragment read is one ds_read_b128 per lane at
  // row (lane & 31), half (lane >> 5).  ds_read_b128 is serviced in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32
  // (MI355X_MICROARCH.md, LDS): with half-0 data always in the even 16-byte slots a group touches only 8 of the 16 slots of a
  // 256-byte bank row (2-way conflict on every fragment read; SQ_LDS_BANK_CONFLICT was 13 % of the kernel's cycles).  Swapping
  // the halves of rows with bit 3 set (rows r and r + 8 share a group) makes the 16 slots of every group distinct.
  const int tf_wr = tf_tile * 8 + ((2 * tf_cp) ^ (((tf_tile >> 3) & 1) << 2));
  f32x2 d[16], t[16];
  auto tf_load = [&](int buf) {
    const float *R = raw_lds + buf * WG_RAW_FLOATS + tf_rd;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) d[r * 4 + q] = *reinterpret_cast<const f32x2 *>(R + (r * RC + q) * 8);
  };
  auto tf_rows = [&]() {  // t = B^T d
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      t[0 + q] = d[0 + q] - d[8 + q];
      t[4 + q] = d[4 + q] + d[8 + q];
      t[8 + q] = d[8 + q] - d[4 + q];
      t[12 + q] = d[4 + q] - d[12 + q];
    }
  };
  auto tf_cols_store = [&](int xi, int buf) {  // V[xi][.] = t[xi][.] B  -> components xi*4 .. xi*4+3
    float *Vw = v_lds + buf * WG_V_FLOATS + tf_wr + xi * 4 * 512;
    *reinterpret_cast<f32x2 *>(Vw + 0 * 512) = t[xi * 4 + 0] - t[xi * 4 + 2];
    *reinterpret_cast<f32x2 *>(Vw + 1 * 512) = t[xi * 4 + 1] + t[xi * 4 + 2];
    *reinterpret_cast<f32x2 *>(Vw + 2 * 512) = t[xi * 4 + 2] - t[xi * 4 + 1];
    *reinterpret_cast<f32x2 *>(Vw + 3 * 512) = t[xi * 4 + 1] - t[xi * 4 + 3];
  };

  // the same work cut into single-shadow items for the loop schedule below
  auto tf_load_item = [&](int buf, int i) {  // two adjacent pixels of one patch row: one ds_read2_b64
    const float *R = raw_lds + buf * WG_RAW_FLOATS + tf_rd + ((i >> 1) * RC + (i & 1) * 2) * 8;
    d[2 * i] = *reinterpret_cast<const f32x2 *>(R);
    d[2 * i + 1] = *reinterpret_cast<const f32x2 *>(R + 8);
  };
  auto tf_rows_item = [&](int q) {  // column q of t = B^T d: 4 packed adds
    t[0 + q] = d[0 + q] - d[8 + q];
    t[4 + q] = d[4 + q] + d[8 + q];
    t[8 + q] = d[8 + q] - d[4 + q];
    t[12 + q] = d[4 + q] - d[12 + q];
  };
  auto tf_cols_compute = [&](int xi) {  // V[xi][.] = t[xi][.] B, in place of t[xi][.]
    const f32x2 v0 = t[xi * 4 + 0] - t[xi * 4 + 2], v1 = t[xi * 4 + 1] + t[xi * 4 + 2];
    const f32x2 v2 = t[xi * 4 + 2] - t[xi * 4 + 1], v3 = t[xi * 4 + 1] - t[xi * 4 + 3];
    t[xi * 4 + 0] = v0; t[xi * 4 + 1] = v1; t[xi * 4 + 2] = v2; t[xi * 4 + 3] = v3;
  };
  auto tf_store_item = [&](int i, int buf) {  // components 2i, 2i+1: one ds_write2st64_b64
    float *Vw = v_lds + buf * WG_V_FLOATS + tf_wr + (2 * i) * 512;
    *reinterpret_cast<f32x2 *>(Vw) = t[2 * i];
    *reinterpret_cast<f32x2 *>(Vw + 512) = t[2 * i + 1];
  };

  const int c0 = split * cps;
  const int c1 = min(a.nchunks, c0 + cps);
  // prologue: one DMA round trip for both raw tiles and the first weight slices (the accumulator zeroing
  // overlaps it), then the first input transform
  const int n_more = c1 - 1 - c0;  // chunks that prefetch a successor
  const int b0 = n_more & 1;       // chunk c lives in buffer (c - c0 + b0) & 1, so that the LAST chunk is always in buffer 0
  issue_raw(c0, b0);
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_u(c0, b0, i);
  issue_raw(min(c0 + 1, c1 - 1), b0 ^ 1);
  f32x16 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
  // ABL bit 7 (round 5, timing only, garbage results): the prologue's DMA round trip is NOT waited for — what a persistent launch that
  // prefetched the next tile's first chunk under the previous tile's epilogue could hide at the very best (tools/ablate_wino_prologue.py)
  if constexpr ((ABL & 128) == 0) dma_wait_all();
  __syncthreads();
  tf_load(b0);
  tf_rows();
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) tf_cols_store(xi, b0);
  __syncthreads();

  const int frag_sw = (half ^ ((l31 >> 3) & 1)) * 4;  // the bank swizzle (mbase / nbase are multiples of 32)
  const int frag_u = (mbase + l31) * 8 + frag_sw, frag_v = (nbase + l31) * 8 + frag_sw;
  // Loop-body scheduling (from the ISA and an s_memtime trace of the loop, tools/wino_trace.py):
  //  * a wave issues in order: an instruction placed between two MFMAs runs in the first one's 64-cycle shadow, but
  //    anything beyond ~60 cycles of issue time in one shadow delays the matrix pipe (a pair with 7 DMA issues + 8
  //    ds_reads behind its first MFMAs measured 980 cycles instead of 512; one with 4 packed adds + 2 ds_writes in one
  //    shadow 665).  So the per-chunk side work is cut into 31 single-shadow items (11 DMA issues, 8 ds_read2 of the
  //    raw patch, 4 row-transform items, 8 column-transform + ds_write2 items) and dealt one per MFMA shadow;
  //  * the compiler forces every lgkmcnt wait that follows a global_load_lds to lgkmcnt(0) (LDS-DMA is a FLAT op),
  //    so each pair issues its first MFMA before the NEXT pair's fragment loads (the forced wait then only covers
  //    loads issued 7 MFMAs earlier) and LDS items are kept out of a pair's last two shadows;
  //  * branch-free: MORE is a compile-time tag (the last chunk runs the no-prefetch copy);
  //  * the chunk barrier sits before the last pair's MFMAs: every LDS read of the chunk has been issued and waited
  //    for by then, and the next chunk's first fragments are fetched under pair 7's 8 MFMAs.
  f32x4 af[2][2], bf[2][2];
  auto load_frags = [&](int buf, int pr, int slot) {
    const float *Ul = u_lds + buf * WG_U_FLOATS + frag_u, *Vl = v_lds + buf * WG_V_FLOATS + frag_v;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      af[slot][k] = *reinterpret_cast<const f32x4 *>(Ul + (2 * pr + k) * 512);
      bf[slot][k] = *reinterpret_cast<const f32x4 *>(Vl + (2 * pr + k) * 512);
    }
  };
  load_frags(b0, 0, 0);
  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  auto body = [&](int c, auto more_tag, auto parity_tag) {  // buffer parity s is a compile-time tag: LDS offsets become immediates
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int s = decltype(parity_tag)::value;
#pragma unroll
    for (int p = 0; p < 8; ++p) {  // component pair (2p, 2p+1): two independent accumulator chains
      const int cur = p & 1;
      if constexpr ((ABL & 64) != 0) {
        if (p == 0) tr0 = __builtin_amdgcn_s_memtime();
        if (a.trace && blockIdx.x < 4 && blockIdx.y == 0 && tid == 0 && (c - c0) == 5) a.trace[4 * 64 * 4 + blockIdx.x * 8 + p] = __builtin_amdgcn_s_memtime();
      }
      if (p == 7 && MORE) {
        if constexpr ((ABL & 64) != 0) tr1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's last reads of U[s]/V[s] are done before others may overwrite them
        if constexpr ((ABL & 64) != 0) tr2 = __builtin_amdgcn_s_memtime();
        dma_wait_all();  // the next chunk's weight slices / raw tile issued by this wave have landed
        if constexpr (!(ABL & 2)) __syncthreads();
        if constexpr ((ABL & 64) != 0) {
          const unsigned long long tr3 = __builtin_amdgcn_s_memtime();
          if (a.trace && blockIdx.x < 4 && blockIdx.y == 0 && tid == 0) {
            unsigned long long *o = a.trace + ((size_t)blockIdx.x * 64 + (c - c0)) * 4;
            o[0] = tr0; o[1] = tr1 - tr0; o[2] = tr2 - tr1; o[3] = tr3 - tr2;
          }
        }
        if constexpr (!(ABL & 16)) load_frags(s ^ 1, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // MFMA i of the pair: accumulator 2p + (i&1), k-pair i>>1
        __builtin_amdgcn_sched_barrier(0);
        acc[2 * p + (i & 1)] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i & 1][i >> 1], bf[cur][i & 1][i >> 1], acc[2 * p + (i & 1)], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0 && p + 1 < 8 && !(ABL & 16)) load_frags(s, p + 1, cur ^ 1);
        if constexpr (MORE) {
          // side work: LDS / DMA instructions are free in an MFMA shadow (up to LDS bandwidth) and are dealt one per
          // shadow; VALU instructions are NOT (they cost their issue time plus a ~20-cycle bubble each time the MFMA
          // stream is broken), so the whole input transform's arithmetic sits in ONE cluster
          const int m = p * 8 + i;
          if constexpr (!(ABL & 1)) {
            if (m >= 1 && m <= 8) issue_u(c + 1, s ^ 1, m - 1);
            else if (m >= 9 && m <= 11) issue_raw_item(min(c + 2, c1 - 1), s, m - 9);  // past the end: a harmless re-load of the last chunk
          }
          if constexpr (!(ABL & 4)) {
            if (m >= 12 && m <= 19) tf_load_item(s ^ 1, m - 12);
            else if (m == 27) {
#pragma unroll
              for (int q = 0; q < 4; ++q) tf_rows_item(q);
#pragma unroll
              for (int xi = 0; xi < 4; ++xi) tf_cols_compute(xi);
            } else if (m >= 28 && m <= 35) tf_store_item(m - 28, s ^ 1);
          }
        }
      }
    }
  };
  if constexpr ((ABL & 64) != 0) tk1 = __builtin_amdgcn_s_memtime();
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int cc = c0;
  if (n_more & 1) { body(cc, std::true_type{}, P1{}); ++cc; }   // the first chunk sits in buffer n_more & 1, the last in buffer 0
  for (; cc < c1 - 1; cc += 2) { body(cc, std::true_type{}, P0{}); body(cc + 1, std::true_type{}, P1{}); }
  // the epilogue's bias vectors are fetched under the last chunk's MFMAs (the transform registers are free there): a
  // global load inside the epilogue would sit behind an s_waitcnt vmcnt(0) that also waits for the previous channel
  // group's STORES to be acknowledged (measured: 9.8k-cycle epilogue)
  f32x4 bias4[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int cb = min((cout0 + mbase) / 8 + g, a.out_cb - 1);
    bias4[g] = *reinterpret_cast<const f32x4 *>(a.bpk + cb * 8 + half * 4);
  }
  body(c1 - 1, std::false_type{}, P0{});
  if constexpr ((ABL & 64) != 0) tk2 = __builtin_amdgcn_s_memtime();

  // ---- output transform A^T M A (register-local), bias, ReLU, stores, fused 2x2 max-pool
  if constexpr ((ABL & 8) != 0) {  // timing experiment: no output transform / stores (the never-true store keeps the accumulators live)
    if (a.H < 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x16 *>(a.part + (size_t)k * 16 + lane * 256) = acc[k];
    }
    return;
  }
  // VALU time is fully exposed here (nothing left to overlap it with), so the epilogue is written for instruction count:
  // two accumulator registers per packed op (v_pk_add_f32 / v_pk_max_f32), one wave-uniform 64-bit base per channel block
  // and 32-bit per-lane byte offsets for the stores (no 64-bit VALU address arithmetic).
  const int tau = nbase + l31;
  const int y = y0 + 2 * (tau / TC), x = x0 + 2 * (tau % TC);
  float *const obase = partial ? a.part + (size_t)split * a.part_slab : a.out;
  const unsigned off00 = (unsigned)((((y + 1) * a.out_Wp + x + 1) * 8 + half * 4) * 4);  // byte offset of output pixel (y, x) in a plane
  const unsigned row_b = (unsigned)(a.out_Wp * 32);
  const bool okx1 = x + 1 < a.W, oky1 = y + 1 < a.H, ok00 = y < a.H && x < a.W;
  const bool okk[4] = {ok00, ok00 && okx1, ok00 && oky1, ok00 && okx1 && oky1};
  const unsigned offk[4] = {off00, off00 + 32u, off00 + row_b, off00 + row_b + 32u};
  const int py = y >> 1, px = x >> 1;
  const bool okp = a.pool && py < a.pool_H && px < a.pool_W;
  const unsigned offp = (unsigned)((((py + 1) * a.pool_Wp + px + 1) * 8 + half * 4) * 4);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int cb = (cout0 + mbase) / 8 + g;  // wave-uniform
    if (cb >= a.out_cb) continue;
    f32x2 Y[4][2];  // [output pixel k][element pair]
#pragma unroll
    for (int ep = 0; ep < 2; ++ep) {
      const int r = g * 4 + ep * 2;
      f32x2 m[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = f32x2{acc[k][r], acc[k][r + 1]};
      const f32x2 s0 = m[0] + m[4] + m[8], s1 = m[1] + m[5] + m[9], s2 = m[2] + m[6] + m[10], s3 = m[3] + m[7] + m[11];
      // (the compiler scalarises a packed fsub whose operands come out of accumulator registers: spell it as the instruction)
      const f32x2 u0 = pk_sub(pk_sub(m[4], m[8]), m[12]), u1 = pk_sub(pk_sub(m[5], m[9]), m[13]);
      const f32x2 u2 = pk_sub(pk_sub(m[6], m[10]), m[14]), u3 = pk_sub(pk_sub(m[7], m[11]), m[15]);
      Y[0][ep] = s0 + s1 + s2; Y[1][ep] = pk_sub(pk_sub(s1, s2), s3);
      Y[2][ep] = u0 + u1 + u2; Y[3][ep] = pk_sub(pk_sub(u1, u2), u3);
    }
    if (partial) {  // raw partial sums; conv_splitk_reduce_kernel finishes the layer (or the layer's tail tiles)
      char *const pb = reinterpret_cast<char *>(obase + (size_t)cb * a.out_plane);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (okk[k]) *reinterpret_cast<f32x4 *>(pb + offk[k]) = f32x4{Y[k][0][0], Y[k][0][1], Y[k][1][0], Y[k][1][1]};
      continue;
    }
    const f32x2 b01 = f32x2{bias4[g][0], bias4[g][1]}, b23 = f32x2{bias4[g][2], bias4[g][3]};
    const f32x2 zero2 = f32x2{0.0f, 0.0f};
    f32x2 mx0 = f32x2{-INFINITY, -INFINITY}, mx1 = mx0;
    char *const ob = reinterpret_cast<char *>(a.out + (size_t)cb * a.out_plane);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x2 v0 = Y[k][0] + b01, v1 = Y[k][1] + b23;
      if (a.relu) { v0 = __builtin_elementwise_max(v0, zero2); v1 = __builtin_elementwise_max(v1, zero2); }
      if (okk[k] && a.out) *reinterpret_cast<f32x4 *>(ob + offk[k]) = f32x4{v0[0], v0[1], v1[0], v1[1]};
      if (okk[k]) { mx0 = __builtin_elementwise_max(mx0, v0); mx1 = __builtin_elementwise_max(mx1, v1); }
    }
    if (okp) *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(a.pool + (size_t)cb * a.pool_plane) + offp) = f32x4{mx0[0], mx0[1], mx1[0], mx1[1]};
  }
  if constexpr ((ABL & 64) != 0) {
    const unsigned long long tk3 = __builtin_amdgcn_s_memtime();
    if (a.trace && blockIdx.x < 4 && blockIdx.y == 0 && tid == 0) {
      unsigned long long *o = a.trace + 4 * 64 * 4 + 32 + blockIdx.x * 4;
      o[0] = tk1 - tk0; o[1] = tk2 - tk1; o[2] = tk3 - tk2; o[3] = tk0;
    }
    if (a.trace && blockIdx.y == 0 && tid == 0 && blockIdx.x < 4096) {  // every block: which CU, when it started / ended
      unsigned long long *o = a.trace + 4 * 64 * 4 + 48 + (size_t)blockIdx.x * 3;
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      o[0] = ((unsigned long long)xcc << 32) | hw; o[1] = tk0; o[2] = tk3;
    }
  }
}

template <int ABL>
static int launch_conv_wino_t(const ConvArgs &a, int tiles_y, hipStream_t s) {
  auto kern = a.wino_tc == 16 ? conv3x3_wino_kernel<ABL, 16> : conv3x3_wino_kernel<ABL, 8>;
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(kern), (int)WG_LDS_BYTES); if (rc_attr) return rc_attr; }
  const int blocks = a.n_ct * tiles_y * a.tiles_x;
  dim3 grid((unsigned)(a.tail_splits > 1 ? a.tail_first + (blocks - a.tail_first) * a.tail_splits : blocks), (unsigned)a.splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), WG_LDS_BYTES, s, a);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

static int launch_conv_wino(const ConvArgs &a, int tiles_y, hipStream_t s) {
#ifdef MPN_DEBUG_HOOKS
  switch (a.ablate) {  // timing experiments only (wrong results); debug flavour of the library only
    case 1: return launch_conv_wino_t<1>(a, tiles_y, s);
    case 2: return launch_conv_wino_t<2>(a, tiles_y, s);
    case 4: return launch_conv_wino_t<4>(a, tiles_y, s);
    case 8: return launch_conv_wino_t<8>(a, tiles_y, s);
    case 5: return launch_conv_wino_t<5>(a, tiles_y, s);
    case 23: return launch_conv_wino_t<23>(a, tiles_y, s);
    case 31: return launch_conv_wino_t<31>(a, tiles_y, s);
    case 64: return launch_conv_wino_t<64>(a, tiles_y, s);
    case 128: return launch_conv_wino_t<128>(a, tiles_y, s);
    case 136: return launch_conv_wino_t<136>(a, tiles_y, s);
    default: break;
  }
#endif
  return launch_conv_wino_t<0>(a, tiles_y, s);
}

// split-K finish: sums S partial slabs in split order (deterministic), + bias, ReLU; writes the C8P
// output and/or its ceil-mode 2x2 max-pool.  One thread per (channel block, pooled-or-full pixel, half).
__global__ void conv_splitk_reduce_kernel(const float *__restrict__ part, size_t slab, int S, size_t plane, int Wp, int H, int W,
                                          int out_cb, const float *__restrict__ bpk, int relu, float *__restrict__ out,
                                          float *__restrict__ pool, size_t pool_plane, int pool_Wp, int pool_H, int pool_W,
                                          int tile_first, int tiles_x, int g_start, int th_log, int tw_log) {
  // tile_first > 0 (Winograd tail split): only the (1 << th_log) x (1 << tw_log)-pixel tiles with index >= tile_first carry
  // partial slabs; the grid covers output rows g_start .. only
  const bool pooling = pool != nullptr;
  const int GHf = pooling ? pool_H : H, GW = pooling ? pool_W : W;
  const int GH = GHf - g_start;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)out_cb * GH * GW * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int gx = (int)(r % GW); r /= GW;
  const int gy = g_start + (int)(r % GH); const int cb = (int)(r / GH);
  if (tile_first > 0) {
    const int ty = (pooling ? 2 * gy : gy) >> th_log, tx = (pooling ? 2 * gx : gx) >> tw_log;
    if (ty * tiles_x + tx < tile_first) return;
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bpk + cb * 8 + h * 4);
  f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  const int ny = pooling ? 2 : 1, nx = pooling ? 2 : 1;
  for (int dy = 0; dy < ny; ++dy)
    for (int dx = 0; dx < nx; ++dx) {
      const int y = pooling ? 2 * gy + dy : gy, x = pooling ? 2 * gx + dx : gx;
      if (y >= H || x >= W) continue;
      const size_t off = (size_t)cb * plane + ((size_t)(y + 1) * Wp + x + 1) * 8 + h * 4;
      f32x4 v = *reinterpret_cast<const f32x4 *>(part + off);
      for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4 *>(part + s * slab + off);
      v += b4;
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] < 0.0f ? 0.0f : v[e];
      }
      if (out) *reinterpret_cast<f32x4 *>(out + off) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
    }
  if (pooling) *reinterpret_cast<f32x4 *>(pool + (size_t)cb * pool_plane + ((size_t)(gy + 1) * pool_Wp + gx + 1) * 8 + h * 4) = m;
}

MPN_KNOB(unsigned long long *, g_wino_trace, nullptr);  // tools/wino_trace.py
MPN_KNOB(int, g_conv_split, 0);          // 0 = auto, >0 = force this many splits (test/bench hook)

// Fill model: 256 CUs, one 128x4 (or two 64x8) blocks resident per CU -> a launch of b equal blocks takes
// ceil(b/slots) rounds.  Split K when that lifts the fill by a margin that pays for the extra slab pass.
static int conv_pick_splits(int blocks, int nchunks, int slots) {
  if (g_conv_split > 0) return g_conv_split < nchunks ? g_conv_split : nchunks;
  if (g_conv_split < 0) return 1;
  auto fill = [&](int b) { return (double)b / ((double)slots * ((b + slots - 1) / slots)); };
  int best = 1;
  double best_score = fill(blocks);
  for (int S = 2; S <= 8; ++S) {
    if (nchunks / S < 6) break;
    double score = fill(blocks * S) * (1.0 - 0.04 - 0.01 * S);
    if (score > best_score * 1.08) { best_score = score; best = S; }
  }
  return best;
}

// Split-K choice for the Winograd kernel from its measured cost model (tools/wino_trace.py): one block = 4.74 k cycles per
// 8-channel chunk + 14 k cycles of prologue / epilogue, 256 blocks per round; a split layer pays the slab pass
// ((S + 1) x output bytes at ~8 TB/s — the slabs are still in the 256-MB Infinity Cache — + ~3 us for the extra launch;
// calibrated against tools/wino_split_sweep.py).
static int wino_pick_splits(int blocks, int nchunks, size_t out_bytes) {
  if (g_conv_split > 0) return g_conv_split < nchunks ? g_conv_split : nchunks;
  if (g_conv_split < 0) return 1;
  const double t_chunk = 4740.0, t_ovh = 14000.0, hz = 2.35e9;
  int best = 1;
  double best_t = 1e30;
  for (int S = 1; S <= 8 && S <= nchunks; ++S) {
    const int cps = (nchunks + S - 1) / S;
    const int Se = (nchunks + cps - 1) / cps;
    if (Se != S) continue;
    const long long rounds = ((long long)blocks * S + 255) / 256;
    double t = rounds * (cps * t_chunk + t_ovh);
    if (S > 1) t += ((S + 1) * (double)out_bytes / 8.0e12 + 3e-6) * hz;
    if (t < best_t * 0.995) { best_t = t; best = S; }
  }
  return best;
}

// Uniform split-K or a TAIL split (ConvArgs::tail_*), whichever the same cost model rates faster.  A launch of B equal blocks
// on 256 one-block CUs runs floor(B / 256) full rounds and then a round in which only B % 256 CUs work for a whole block
// time; cutting just those tiles' K range in S pieces makes that last round ~1/S as long (VGG conv3_x at 150 x 250: 640
// blocks = 2.5 rounds -> 3 block times before, 2 + 0.5 after), for one small reduce over the tail tiles only.
struct WinoPlan { int splits, tail_first, tail_splits, tail_cps; };
static WinoPlan wino_pick_plan(int blocks, int n_ct, int nchunks, size_t out_bytes) {
  WinoPlan p{1, 0, 0, 0};
  if (g_conv_split < 0) {  // test hook: force a tail split of -g_conv_split over the second half of the tiles
    const int S = -g_conv_split < nchunks ? -g_conv_split : nchunks;
    const int first = (blocks / 2 / n_ct) * n_ct;
    if (S > 1 && first > 0 && first < blocks) {
      p.tail_cps = cdiv(nchunks, S); p.tail_splits = cdiv(nchunks, p.tail_cps); p.tail_first = first;
      if (p.tail_splits < 2) p = WinoPlan{1, 0, 0, 0};
    }
    return p;
  }
  p.splits = wino_pick_splits(blocks, nchunks, out_bytes);
  if (g_conv_split > 0) return p;
  const double t_chunk = 4740.0, t_ovh = 14000.0, hz = 2.35e9;
  auto t_uniform = [&](int S) {
    const int cps = cdiv(nchunks, S);
    const long long rounds = ((long long)blocks * cdiv(nchunks, cps) + 255) / 256;
    double t = rounds * (cps * t_chunk + t_ovh);
    if (S > 1) t += ((S + 1) * (double)out_bytes / 8.0e12 + 3e-6) * hz;
    return t;
  };
  double best_t = t_uniform(p.splits);
  const int full = (blocks / 256) * 256 / n_ct * n_ct, rem = blocks - full;
  if (full > 0 && rem > 0) {
    const double tile_bytes = 64.0 * 256.0 * 4.0;  // one block's outputs
    for (int S = 2; S <= 8 && S <= nchunks / 2; ++S) {
      const int cps = cdiv(nchunks, S), Se = cdiv(nchunks, cps);
      if (Se != S) continue;
      const long long rounds = ((long long)rem * S + 255) / 256;
      const double t = (double)(full / 256) * (nchunks * t_chunk + t_ovh) + rounds * (cps * t_chunk + t_ovh) +
                       ((S + 1) * rem * tile_bytes / 8.0e12 + 3e-6) * hz;
      if (t < best_t * 0.98) { best_t = t; p = WinoPlan{1, full, S, cps}; }
    }
  }
  return p;
}

MPN_KNOB(int, g_wino_tc, 0);  // test hook: force the Winograd block geometry (8 / 16; 0 = per layer)
MPN_KNOB(int, g_conv_variant, 0);  // 0 = auto; test/bench hook: 1 = 128x4 tile / 9 taps per stage, 2 = 64x8 / 9, 3 = 128x4 / 3, 4 = 64x8 / 3,
                                 // 5 = 128 couts x 8 rows (64x128 per wave, 8 accumulators), 6 = 64 couts x 16 rows

int conv3x3_variant_for(int Cout, bool has_wino) {
  int variant = g_conv_variant & 15;
  // measured on MI355X (tools/bench_layers.py): Winograd F(2x2,3x3) beats the direct kernels on every VGG layer with
  // >= 16 input channels (2.33 vs 3.53 ms for the trunk); among the direct kernels one 4-wave block per CU (9 taps per
  // stage) beats the 3-blocks-per-CU variants — co-resident waves only time-share the SIMD's matrix pipe.
  if (variant == 7 && !has_wino) variant = 0;
  if (variant == 0) variant = has_wino ? 7 : ((Cout <= 64) ? 2 : 1);
  return variant;
}

int conv3x3_c8p(Act in, const float *d_wpk, const float *d_bpk, int Cout, int relu, Act out, Act pooled, hipStream_t s, const float *d_wino,
                bool batch_invariant) {
  MPN_CHECK_ARG(in.p && (d_wpk || d_wino) && d_bpk && (out.p || pooled.p));
  ConvArgs a{};
  a.in = in.p; a.in_plane = in.plane(); a.in_Wp = in.Wp;
  a.wpk = d_wpk; a.CoutP = conv_coutp(Cout); a.bpk = d_bpk;
  a.out = out.p; a.out_plane = out.p ? out.plane() : 0; a.out_Wp = out.p ? out.Wp : 0;
  a.pool = pooled.p; a.pool_plane = pooled.p ? pooled.plane() : 0; a.pool_Wp = pooled.p ? pooled.Wp : 0;
  a.pool_H = pooled.p ? pooled.H : 0; a.pool_W = pooled.p ? pooled.W : 0;
  a.H = in.H; a.W = in.W; a.nchunks = in.Cb(); a.out_cb = (Cout + 7) / 8; a.relu = relu;
  a.tiles_x = cdiv(in.W, 32);
  a.ablate = g_gemm_ablate;
  a.trace = g_wino_trace;
  if (out.p) MPN_CHECK_ARG(out.H == in.H && out.W == in.W && out.C == Cout);
  if (pooled.p) MPN_CHECK_ARG(pooled.H == (in.H + 1) / 2 && pooled.W == (in.W + 1) / 2 && pooled.C == Cout);
  int variant = conv3x3_variant_for(Cout, d_wino != nullptr);
  if (!d_wpk) variant = 7;
  if (variant == 7) {  // Winograd F(2x2,3x3): 64 couts x 16x16 px per block
    a.wpk = d_wino;
    // block geometry: 16 x 16 px or 8 x 32 px of outputs, whichever pads this map less
    const long px_sq = (long)cdiv(in.H, 16) * 16 * cdiv(in.W, 16) * 16, px_wide = (long)cdiv(in.H, 8) * 8 * cdiv(in.W, 32) * 32;
    a.wino_tc = batch_invariant ? 16 : (g_wino_tc == 8 || g_wino_tc == 16) ? g_wino_tc : (px_wide < px_sq ? 16 : 8);
    const int tpx_h = a.wino_tc == 16 ? 8 : 16, tpx_w = a.wino_tc == 16 ? 32 : 16;
    a.tiles_x = cdiv(in.W, tpx_w);
    const int tiles_y = cdiv(in.H, tpx_h);
    a.n_ct = cdiv(Cout, 64);
    const int blocks = a.n_ct * tiles_y * a.tiles_x;
    Act geo = out.p ? out : make_act(nullptr, Cout, in.H, in.W);
    const WinoPlan plan = batch_invariant ? WinoPlan{1, 0, 0, 0} : wino_pick_plan(blocks, a.n_ct, a.nchunks, geo.elems() * sizeof(float));
    a.splits = plan.splits;
    a.chunks_per_split = cdiv(a.nchunks, a.splits);
    a.splits = cdiv(a.nchunks, a.chunks_per_split);
    a.tail_first = plan.tail_first; a.tail_splits = plan.tail_splits; a.tail_cps = plan.tail_cps;
    const int n_slabs = a.tail_splits > 1 ? a.tail_splits : a.splits;
    if (n_slabs > 1) {
      a.part_slab = geo.elems();
      a.out_plane = geo.plane(); a.out_Wp = geo.Wp;
      size_t need = a.part_slab * n_slabs * sizeof(float);
      void *ws = nullptr;
      { int rc_ws = scratch_get(SCR_CONV_SPLITK, need, s, &ws); if (rc_ws) return rc_ws; }
      a.part = static_cast<float *>(ws);
    }
    int rc = launch_conv_wino(a, tiles_y, s);
    if (rc != MPN_OK || n_slabs == 1) return rc;
    const int GH = pooled.p ? pooled.H : in.H, GW = pooled.p ? pooled.W : in.W;
    int tile_first = 0, g_start = 0;
    if (a.tail_splits > 1) {  // only the tail tiles have partial slabs
      tile_first = a.tail_first / a.n_ct;
      const int y_start = (tile_first / a.tiles_x) * tpx_h;
      g_start = pooled.p ? y_start / 2 : y_start;
    }
    const size_t total = (size_t)a.out_cb * (GH - g_start) * GW * 2;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, a.part, a.part_slab, n_slabs, geo.plane(),
                       geo.Wp, in.H, in.W, a.out_cb, d_bpk, relu, out.p, pooled.p, a.pool_plane, a.pool_Wp, a.pool_H, a.pool_W, tile_first,
                       a.tiles_x, g_start, a.wino_tc == 16 ? 3 : 4, a.wino_tc == 16 ? 5 : 4);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  const bool wide = (variant == 1 || variant == 3 || variant == 5);  // 128-cout tiles; else 64-cout tiles
  const int th = variant == 5 ? 8 : (variant == 6 ? 16 : (wide ? 4 : 8));
  const int tiles_y = cdiv(in.H, th);
  a.n_ct = cdiv(Cout, wide ? 128 : 64);
  const int blocks = a.n_ct * tiles_y * a.tiles_x;
  const int slots = (variant == 1 || variant >= 5) ? 256 : (variant == 2 ? 512 : 768);  // co-resident blocks on 256 CUs (LDS / VGPR bound)
  a.splits = conv_pick_splits(blocks, a.nchunks, slots);
  a.chunks_per_split = cdiv(a.nchunks, a.splits);
  a.splits = cdiv(a.nchunks, a.chunks_per_split);
  Act geo = out.p ? out : make_act(nullptr, Cout, in.H, in.W);  // partial slabs use the full-resolution output geometry
  if (a.splits > 1) {
    a.part_slab = geo.elems();
    a.out_plane = geo.plane(); a.out_Wp = geo.Wp;
    size_t need = a.part_slab * a.splits * sizeof(float);
    void *ws = nullptr;
    { int rc_ws = scratch_get(SCR_CONV_SPLITK, need, s, &ws); if (rc_ws) return rc_ws; }
    a.part = static_cast<float *>(ws);
  }
  int rc;
  switch (variant) {
    case 1: rc = launch_conv<128, 4, 2, 2, 9>(a, tiles_y, s); break;
    case 2: rc = launch_conv<64, 8, 1, 4, 9>(a, tiles_y, s); break;
    case 3: rc = launch_conv<128, 4, 2, 2, 3>(a, tiles_y, s); break;
    case 5: rc = launch_conv<128, 8, 2, 2, 9>(a, tiles_y, s); break;
    case 6: rc = launch_conv<64, 16, 1, 4, 9>(a, tiles_y, s); break;
    default: rc = launch_conv<64, 8, 1, 4, 3>(a, tiles_y, s); break;
  }
  if (rc != MPN_OK || a.splits == 1) return rc;
  const int GH = pooled.p ? pooled.H : in.H, GW = pooled.p ? pooled.W : in.W;
  const size_t total = (size_t)a.out_cb * GH * GW * 2;
  hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, a.part, a.part_slab, a.splits, geo.plane(),
                     geo.Wp, in.H, in.W, a.out_cb, d_bpk, relu, out.p, pooled.p, a.pool_plane, a.pool_Wp, a.pool_H, a.pool_W, 0, 0, 0, 4, 4);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// =================================================================================================
// linear: y[M,N] = x[M,K] W[N,K]^T (+b, ReLU) on C8 matrices, deterministic split-K
// =================================================================================================
struct GemmArgs {
  const float *x; int Mp;       // [K8/8][Mp][8]
  const float *wpk; int NP;     // [K8/8][NP][8]
  const float *bpk;
  float *y;                     // [NP/8][Mp][8]   (split: partial slabs [S][NP/8][Mp][8])
  int M, nstages, stages_per_split, relu, n_mt, n_nt, direct;
  int seg_stages;  // un-split launch of a row-invariant GEMM: fold the accumulator into the running total every seg_stages stages (0 = never)
  int n_fast;  // tile order: 0 = all row tiles of one column tile first (the weights are the big operand: fc6), 1 = all column tiles of one
               // row tile first (the activations are: ResNet's pointwise convolutions over 10^5 pixel rows) — the big operand is streamed once
  const float *res;  // optional residual in y's layout, added before the ReLU (direct mode only; ResNet 1x1 convolutions)
  // K segments with a per-ROW scale each (un-split launches only; MultiPathNet's mix GEMM: nn.Normalize of the three pooled maps folded
  // into the accumulator fold instead of a read-modify-write pass over the pooled matrix): y = sum_seg rs[seg][row % rs_mod] * (x_seg . w_seg)
  int nsb, sb0, sb1;             // interior segment boundaries (stage indices), nsb = 0: none (seg_stages applies)
  const float *rs0, *rs1, *rs2;  // scale vector of segment 0 / 1 / 2 (rs0 == nullptr: no scaling)
  int rs_mod;
  // Round 6, MultiPathNet's mix GEMM: its rows are (bin, roi) pairs and its output IS fc6's operand [cout block][bin][Mp6][8].  Until round 5
  // every bin carried Mp6 = N rounded up to 128 rows (49 x 1024 for 1000 proposals: 392 row tiles, 1568 blocks = 3.06 rounds of 512 resident
  // blocks — a whole extra round for 32 blocks).  Now the operand's rows are packed, bin_rows = N rounded up to 8 per bin (49 x 1000 rows: 383
  // row tiles, 1532 blocks = 2.99 rounds), a row tile may straddle bins, and the epilogue scatters row m to (bin m / bin_rows, roi m % bin_rows).
  int xp;                        // row pitch of x between K chunks (Mp unless the rows are packed: the last row tile then reads past the
                                 // chunk's rows — into the next chunk or the buffer's slack — and never stores them)
  int bin_rows, out_Mp;          // bin_rows > 0: the scatter above into [NP/8][M / bin_rows][out_Mp][8]
};

// C8 GEMM  y[NP/8][Mp][8] = x[K/8][Mp][8] . wpk[K/8][NP][8]: 128 x 128 tile per block, KCH 8-wide K chunks per LDS stage (4 -> 32 k, 32 KiB per
// stage), two stages in LDS, operands by LDS-DMA.  Software-pipelined by hand (see the scheduling notes in conv3x3_wino_kernel): branch-free stage body,
// operand fragments double-buffered in registers one K chunk ahead, each chunk's first MFMA issued BEFORE the next
// chunk's fragment loads (so the lgkmcnt(0) the compiler forces after LDS-DMA only covers loads issued 15 MFMAs
// earlier), the next stage's DMA spread behind MFMAs 1-4 of the first chunk, and the stage barrier placed before the
// LAST chunk's MFMAs so the next stage's first fragments are fetched under them.
// FOLD: the launch folds its accumulator into a running total at K-segment boundaries (row-invariant summation / per-row-scaled segments).
// A separate instantiation because the total costs 64 more registers (264 > 256: one block per CU instead of two); launches that never
// fold — split-K pieces, ResNet's pointwise convolutions, the un-scaled mix — keep the two-blocks-per-CU form.
// RSI (round 6; with FOLD = false): per-row-scaled K segments WITHOUT the running total — y = s_last * (P_last + (s_k / s_last) * (... )): at a segment
// boundary the accumulator is multiplied by the ratio of the two segments' row scales and keeps accumulating, at the end by the last segment's
// scale.  Mathematically the FOLD form's sum_k s_k P_k with two or three more roundings per element (2^-24 each, far inside the 1e-4 of the
// path's parity gate), for 64 fewer registers: 196 + 8 instead of 264 — two blocks per CU, and, what matters more on the MultiPathNet path, a
// mix-GEMM block can share a CU with a block of the other tower lane's fc6 / fc7 (264 registers per lane: 264 + 264 > 512, 204 + 264 fits).
template <int KCH, bool FOLD, bool RSI = false>
__global__ __launch_bounds__(256) void gemm_c8_pf_kernel(GemmArgs a) {
  static_assert(!(FOLD && RSI), "row scales are applied either through the running total or in place");
  constexpr int OP_FLOATS = KCH * 128 * 8;
  constexpr int STAGE = 2 * OP_FLOATS;
  constexpr int IT = OP_FLOATS / 256 / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int b = blockIdx.x;
  const int nb = gridDim.x;
  {  // the blocks of one XCD (b mod 8, round-robin dispatch) walk neighbouring tiles: XCD x owns nb / 8 (+ 1 for x < nb mod 8) consecutive ones.
     // (Until round 6 only launches of a multiple of 8 blocks were remapped — the packed mix GEMM's 1532 blocks fetched every row tile into four L2s.)
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7, idx = b >> 3;
    b = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
  }
  const int nt = a.n_fast ? b % a.n_nt : b / a.n_mt, mt = a.n_fast ? b / a.n_nt : b - nt * a.n_mt;
  const int n0 = nt * 128, m0 = mt * 128;
  const int split = blockIdx.y;
  const int st0 = split * a.stages_per_split;
  const int st1 = min(a.nstages, st0 + a.stages_per_split);
  const int wm = wave >> 1, wn = wave & 1;

  // DMA item i of this wave = K chunk i of the stage (uniform), 16-byte piece wave*64 + lane of its 128 rows: SADDR form,
  // no VALU address arithmetic in the loop (see glds16_saddr)
  static_assert(IT == KCH, "one 1-KiB wave-load per wave per K chunk per operand");
  const unsigned dma_lane = (unsigned)((wave * 64 + lane) * 16);
  const size_t a_stage = (size_t)KCH * a.NP * 8, b_stage = (size_t)KCH * a.xp * 8;
  const float *const a_tile = a.wpk + (size_t)n0 * 8, *const b_tile = a.x + (size_t)m0 * 8;
  const unsigned lds0 = lds_byte_addr(lds) + (unsigned)(wave * 256) * 4;
  auto issue_a = [&](int st, int s, int i) {
    glds16_saddr(a_tile + (size_t)st * a_stage + (size_t)i * a.NP * 8, dma_lane, lds0 + (unsigned)(s * STAGE + i * 1024) * 4);
  };
  auto issue_b = [&](int st, int s, int i) {
    glds16_saddr(b_tile + (size_t)st * b_stage + (size_t)i * a.xp * 8, dma_lane, lds0 + (unsigned)(s * STAGE + OP_FLOATS + i * 1024) * 4);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  float rsc[3][2];  // this lane's two rows' scales per K segment (fetched now: a load at a fold boundary would be an exposed round trip)
#pragma unroll
  for (int sg = 0; sg < 3; ++sg) { rsc[sg][0] = 1.0f; rsc[sg][1] = 1.0f; }
  if ((FOLD || RSI) && a.rs0) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int row = (m0 + wn * 64 + ni * 32 + l31) % a.rs_mod;
      rsc[0][ni] = a.rs0[row];
      if (a.rs1) rsc[1][ni] = a.rs1[row];
      if (a.rs2) rsc[2][ni] = a.rs2[row];
    }
    if constexpr (RSI) {  // rsc[k] <- s_k / s_(k+1) for the interior boundaries, the last segment's scale stays: applied at the end
      const int nseg = a.nsb + 1;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const float s0 = rsc[0][ni], s1 = rsc[1][ni], s2 = rsc[2][ni];
        if (nseg == 2) { rsc[0][ni] = s0 / s1; rsc[2][ni] = s1; }
        else if (nseg == 3) { rsc[0][ni] = s0 / s1; rsc[1][ni] = s1 / s2; rsc[2][ni] = s2; }
        else rsc[2][ni] = s0;
      }
    }
  }
  const int lane_off = l31 * 8 + half * 4;
  f32x4 af[2][2], bf[2][2];
  auto load_frags = [&](int s, int kk, int slot) {
    const float *Al = lds + s * STAGE + lane_off, *Bl = Al + OP_FLOATS;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) af[slot][mi] = *reinterpret_cast<const f32x4 *>(Al + (kk * 128 + wm * 64 + mi * 32) * 8);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[slot][ni] = *reinterpret_cast<const f32x4 *>(Bl + (kk * 128 + wn * 64 + ni * 32) * 8);
  };
  if (st0 >= st1) return;  // (never: every split owns at least one stage)
  const int n_more = st1 - 1 - st0;  // stages that prefetch a successor
  const int b0 = n_more & 1;         // stage st lives in buffer (st - st0 + b0) & 1: the LAST stage is always in buffer 0
#pragma unroll
  for (int i = 0; i < IT; ++i) { issue_a(st0, b0, i); issue_b(st0, b0, i); }
  dma_wait_all();
  __syncthreads();
  load_frags(b0, 0, 0);

  static_assert(KCH % 2 == 0, "fragment slot parity assumes an even chunk count per stage");
  auto body = [&](int st, auto more_tag, auto parity_tag) {  // buffer parity is a compile-time tag: LDS offsets become immediates
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int s = decltype(parity_tag)::value;
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
      const int cur = kk & 1;
      if (kk == KCH - 1 && MORE) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's last reads of stage s are done
        dma_wait_all();                      // and the next stage it issued has landed
        __syncthreads();
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
          if (kk == 0 && t >= 1 && t <= IT) { issue_a(st + 1, s ^ 1, t - 1); issue_b(st + 1, s ^ 1, t - 1); }
        }
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // Row-invariant summation (linear_c8's row_invariant): at every canonical segment boundary the accumulator is folded into a
  // running total and restarted from zero — total = (((0 + s_0) + s_1) + ...), the order in which splitk_reduce_kernel adds the
  // slabs when the same GEMM runs one block per segment.  64 VALU adds per wave per boundary against >= 256 MFMAs per segment.
  f32x16 tot[2][2];
  if constexpr (FOLD) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[mi][ni][r] = 0.0f;
  }
  int fseg = 0;  // index of the segment being accumulated
  auto fold = [&]() {
    if constexpr (RSI) {  // boundary fseg -> fseg + 1: the accumulator moves to the next segment's scale
      const float q0 = fseg == 0 ? rsc[0][0] : rsc[1][0], q1 = fseg == 0 ? rsc[0][1] : rsc[1][1];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mi][0][r] *= q0; acc[mi][1][r] *= q1; }
      ++fseg;
      return;
    }
    if constexpr (!FOLD) return;
    if (a.rs0) {  // (wave-uniform) the segment's per-row scale: tot += scale * acc
      const float s0 = fseg == 0 ? rsc[0][0] : (fseg == 1 ? rsc[1][0] : rsc[2][0]), s1 = fseg == 0 ? rsc[0][1] : (fseg == 1 ? rsc[1][1] : rsc[2][1]);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tot[mi][0][r] += s0 * acc[mi][0][r]; acc[mi][0][r] = 0.0f;
          tot[mi][1][r] += s1 * acc[mi][1][r]; acc[mi][1][r] = 0.0f;
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[mi][ni][r] += acc[mi][ni][r]; acc[mi][ni][r] = 0.0f; }
    }
    ++fseg;
  };
  // fold points: explicit boundaries (sb0, sb1) or every seg_stages stages
  int next_fold = !(FOLD || RSI) ? 0x7fffffff : (a.nsb > 0 ? a.sb0 : (a.seg_stages > 0 ? st0 + a.seg_stages : 0x7fffffff));
  auto advance = [&]() { next_fold = a.nsb > 0 ? (fseg < a.nsb ? a.sb1 : 0x7fffffff) : next_fold + a.seg_stages; };
  int st = st0;
  if (n_more & 1) { body(st, std::true_type{}, P1{}); ++st; if (st == next_fold) { fold(); advance(); } }
  for (; st < st1 - 1; st += 2) {
    body(st, std::true_type{}, P0{});
    if (st + 1 == next_fold) { fold(); advance(); }
    body(st + 1, std::true_type{}, P1{});
    if (st + 2 == next_fold) { fold(); advance(); }
  }
  body(st1 - 1, std::false_type{}, P0{});
  if constexpr (RSI) {  // the last segment's scale
    if (a.rs0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[mi][0][r] *= rsc[2][0]; acc[mi][1][r] *= rsc[2][1]; }
    }
  } else
  fold();  // tot = 0 + acc when nothing was folded before: exact

  // Epilogue: every bias / residual load is issued before the first store (the empty asm is a compiler barrier for memory
  // operations).  Stores count in vmcnt on this ISA, so a load that follows a store in program order waits for the store's
  // acknowledgement; interleaved, the 16 (bias, residual) -> store groups of a tile are 16 exposed round trips, and the two
  // blocks that share a CU run in lockstep (launched together, same length), so they reach their epilogues together and the
  // matrix pipe idles for all of it.
  float *yb = a.y + (a.direct ? (size_t)0 : (size_t)split * (a.NP / 8) * a.Mp * 8);
  f32x4 b4[2][4], r4[2][4][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb8 = (n0 + wm * 64 + mi * 32) / 8 + g;
      b4[mi][g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.direct) b4[mi][g] = *reinterpret_cast<const f32x4 *>(a.bpk + nb8 * 8 + half * 4);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int m = m0 + wn * 64 + ni * 32 + l31;
        r4[mi][g][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!FOLD) {  // (a folding launch never carries a residual: 20 fewer registers)
          if (a.res && m < a.M) r4[mi][g][ni] = *reinterpret_cast<const f32x4 *>(a.res + ((size_t)nb8 * a.Mp + m) * 8 + half * 4);
        }
      }
    }
  // where this lane's two rows go: row m of channel block nb8 at (nb8 * cb_rows + row_off) records of 8 — rows as they are, or (bin, roi) scattered
  size_t row_off[2], cb_rows = (size_t)a.Mp;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int m = m0 + wn * 64 + ni * 32 + l31;
    row_off[ni] = (size_t)m;
    if (a.bin_rows > 0) { const int bin = m / a.bin_rows; row_off[ni] = (size_t)bin * a.out_Mp + (m - bin * a.bin_rows); }
  }
  if (a.bin_rows > 0) cb_rows = (size_t)(a.M / a.bin_rows) * a.out_Mp;
  asm volatile("" ::: "memory");
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb8 = (n0 + wm * 64 + mi * 32) / 8 + g;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int m = m0 + wn * 64 + ni * 32 + l31;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (FOLD ? tot[mi][ni][g * 4 + e] : acc[mi][ni][g * 4 + e]) + b4[mi][g][e] + r4[mi][g][ni][e];
          if (a.direct && a.relu) t = t < 0.0f ? 0.0f : t;
          v[e] = t;
        }
        if (m < a.M) *reinterpret_cast<f32x4 *>(yb + ((size_t)nb8 * cb_rows + row_off[ni]) * 8 + half * 4) = v;
      }
    }
}

// sums the split-K slabs in split order (deterministic), adds bias, ReLU; writes C8 and/or row-major
__global__ void splitk_reduce_kernel(const float *__restrict__ part, int S, int NP, int Mp, int M, int N,
                                     const float *__restrict__ bpk, int relu, float *__restrict__ y_c8,
                                     float *__restrict__ y_rm) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [NP/8][M] records of 8
  size_t total = (size_t)(NP / 8) * M;
  if (t >= total) return;
  int nb8 = (int)(t / M), m = (int)(t - (size_t)nb8 * M);
  size_t slab = (size_t)(NP / 8) * Mp * 8, off = ((size_t)nb8 * Mp + m) * 8;
  f32x4 lo = f32x4{0, 0, 0, 0}, hi = lo;
  for (int s = 0; s < S; ++s) {
    lo += *reinterpret_cast<const f32x4 *>(part + s * slab + off);
    hi += *reinterpret_cast<const f32x4 *>(part + s * slab + off + 4);
  }
  lo += *reinterpret_cast<const f32x4 *>(bpk + nb8 * 8);
  hi += *reinterpret_cast<const f32x4 *>(bpk + nb8 * 8 + 4);
  if (relu) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { lo[e] = lo[e] < 0.f ? 0.f : lo[e]; hi[e] = hi[e] < 0.f ? 0.f : hi[e]; }
  }
  if (y_c8) {
    *reinterpret_cast<f32x4 *>(y_c8 + off) = lo;
    *reinterpret_cast<f32x4 *>(y_c8 + off + 4) = hi;
  }
  if (y_rm) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int n = nb8 * 8 + e;
      if (n < N) y_rm[(size_t)m * N + n] = lo[e];
      if (n + 4 < N) y_rm[(size_t)m * N + n + 4] = hi[e];
    }
  }
}

MPN_KNOB(int, g_gemm_kch, 0);       // test/bench hook: force 4 or 8 K chunks per stage
MPN_KNOB(int, g_gemm_split, 0);  // test/bench hook: force a split-K factor

MPN_KNOB(int, g_split3_ranges, 0);  // mpn_debug_set_split3_ranges: forced number of K ranges of linear_c8_split3 (0 = its own rule)
MPN_KNOB(int, g_gemm_rsi, 1);  // mpn_debug_set_gemm_rsi: 0 = per-row-scaled K segments through the running total (the FOLD kernel, rounds 3-5)
static thread_local ScratchSlot t_gemm_splitk_slot = SCR_GEMM_SPLITK;
SplitkSlotScope::SplitkSlotScope(ScratchSlot slot) : prev(t_gemm_splitk_slot) { t_gemm_splitk_slot = slot; }
SplitkSlotScope::~SplitkSlotScope() { t_gemm_splitk_slot = prev; }

bool linear_c8_is_direct(int M, int N, int Mp_override) {
  const int Mp = Mp_override ? Mp_override : lin_mp(M);
  return g_gemm_split == 0 && (Mp / 128) * (lin_np(N) / 128) >= 128;
}

static int linear_c8_impl(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8,
                          float *d_y_rm, hipStream_t s, int Mp_override, const float *d_res_c8, int row_invariant, const GemmRowScale *rs);

int linear_c8(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8,
              float *d_y_rm, hipStream_t s, int Mp_override, const float *d_res_c8, int row_invariant) {
  return linear_c8_impl(d_x_c8, M, K, d_wpk, d_bpk, N, relu, d_y_c8, d_y_rm, s, Mp_override, d_res_c8, row_invariant, nullptr);
}

int linear_c8_rowscaled(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8, hipStream_t s,
                        int Mp_override, const GemmRowScale &rs) {
  MPN_CHECK_ARG(rs.n_seg >= 1 && rs.n_seg <= 3 && rs.rs_mod > 0 && rs.scale[0] != nullptr);
  for (int i = 0; i + 1 < rs.n_seg; ++i) MPN_CHECK_ARG(rs.k_end[i] > (i ? rs.k_end[i - 1] : 0) && rs.k_end[i] % 32 == 0 && rs.k_end[i] < K);
  return linear_c8_impl(d_x_c8, M, K, d_wpk, d_bpk, N, relu, d_y_c8, nullptr, s, Mp_override, nullptr, 0, &rs);
}

static int linear_c8_impl(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8,
                          float *d_y_rm, hipStream_t s, int Mp_override, const float *d_res_c8, int row_invariant, const GemmRowScale *rs) {
  MPN_CHECK_ARG(d_x_c8 && d_wpk && d_bpk && (d_y_c8 || d_y_rm) && M > 0 && K > 0 && N > 0);
  MPN_CHECK_ARG(Mp_override == 0 || (Mp_override >= M && Mp_override % 128 == 0));
  GemmArgs a{};
  a.x = d_x_c8; a.Mp = Mp_override ? Mp_override : lin_mp(M); a.wpk = d_wpk; a.NP = lin_np(N); a.bpk = d_bpk;
  a.M = M; a.relu = relu;
  a.xp = a.Mp;
  if (rs && rs->bin_rows > 0) {  // packed (bin, roi) rows scattered into the consumer's [cout block][bin][out_Mp][8] operand
    MPN_CHECK_ARG(rs->bin_rows % 4 == 0 && M % rs->bin_rows == 0 && rs->out_Mp >= rs->bin_rows && rs->x_pitch >= M);
    a.xp = rs->x_pitch; a.bin_rows = rs->bin_rows; a.out_Mp = rs->out_Mp;
  }
  const int K64 = round_up(K, 64);
  a.n_mt = a.Mp / 128; a.n_nt = a.NP / 128;
  a.n_fast = a.n_mt > a.n_nt ? 1 : 0;
  const int tiles = a.n_mt * a.n_nt;
  // long-K, enough tiles to fill the chip: 64-k stages; otherwise 32-k stages (finer split-K granularity)
  const int kch = g_gemm_kch ? g_gemm_kch : 4;  // 64-k stages measured no faster than 32-k (tools/bench_layers.py)
  a.nstages = K64 / (8 * kch);
  int S = 1, inv_seg = 0;
  a.seg_stages = 0;
  if (rs || row_invariant == 2) S = 1;  // ONE accumulation chain whatever the row count (trivially row-invariant): short-K layers that a
                                        // shard of any size runs with plenty of row tiles — MultiPathNet's mix, the per-ROI pointwise convolutions
  else if (g_gemm_split > 0) S = g_gemm_split;
  else if (row_invariant) {
    // canonical segments from (K, N) alone, >= 4 stages (128 k) each: wide layers (fc6 / fc7, >= 16 column tiles) as many as fill
    // 256 CUs when there is a SINGLE row tile — they only ever run split when a ROI shard is small, which is when it matters —,
    // narrow ones (the cls / bbox / integral heads, always split) as many as fill the chip at the usual 8 row tiles (1000 ROIs):
    // finer segments would only add slab traffic (32 instead of 8 slabs cost the MultiPathNet heads +0.15 ms)
    int Sc = 256 / (a.n_nt * (a.n_nt >= 16 ? 1 : 8));
    if (Sc > 32) Sc = 32;
    if (Sc < 1) Sc = 1;
    int seg = cdiv(a.nstages, Sc);
    if (seg < 4) seg = 4;
    if (tiles < 128) S = cdiv(a.nstages, seg);          // one block per segment + the reduce kernel
    else if (seg < a.nstages) a.seg_stages = seg;       // one block walks the segments, folding at the boundaries
    inv_seg = seg;
  } else if (tiles < 128) {  // too few tiles to fill 256 CUs: split K (deterministic two-pass reduce)
    S = 256 / tiles;
    if (S > a.nstages / 2) S = a.nstages / 2;
    if (S < 1) S = 1;
  }
  a.stages_per_split = (inv_seg && S > 1) ? inv_seg : cdiv(a.nstages, S);  // row-invariant: the segments ARE the canonical ones
  S = cdiv(a.nstages, a.stages_per_split);
  const bool direct = (S == 1) && d_y_c8 && !d_y_rm;
  a.direct = direct ? 1 : 0;
  if (rs) {  // per-row-scaled K segments (checked un-split by the caller): explicit fold boundaries in stages of 8 * kch channels
    if (!direct || kch != 4) { set_error("linear_c8_rowscaled: un-split 32-k-stage launches only"); return MPN_EINVAL; }
    a.nsb = rs->n_seg - 1;
    a.sb0 = rs->n_seg > 1 ? rs->k_end[0] / 32 : 0x7fffffff;
    a.sb1 = rs->n_seg > 2 ? rs->k_end[1] / 32 : 0x7fffffff;
    a.rs0 = rs->scale[0]; a.rs1 = rs->n_seg > 1 ? rs->scale[1] : nullptr; a.rs2 = rs->n_seg > 2 ? rs->scale[2] : nullptr;
    a.rs_mod = rs->rs_mod;
    a.seg_stages = 0;
  }
  if (d_res_c8 && !direct) { set_error("linear_c8: a residual needs the direct (un-split, C8 output) form"); return MPN_EINVAL; }
  a.res = d_res_c8;
  {
    int rc_attr = kch == 8 ? set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_pf_kernel<8, false>), 2 * 2 * 8 * 128 * 8 * 4)
                           : set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_pf_kernel<4, false>), 2 * 2 * 4 * 128 * 8 * 4);
    if (rc_attr == MPN_OK) rc_attr = kch == 8 ? set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_pf_kernel<8, true>), 2 * 2 * 8 * 128 * 8 * 4)
                                             : set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_pf_kernel<4, true>), 2 * 2 * 4 * 128 * 8 * 4);
    if (rc_attr) return rc_attr;
  }
  if (direct) {
    a.y = d_y_c8;
  } else {
    size_t need = (size_t)S * (a.NP / 8) * a.Mp * 8 * sizeof(float);
    void *ws = nullptr;
    { int rc_ws = scratch_get(t_gemm_splitk_slot, need, s, &ws); if (rc_ws) return rc_ws; }
    a.y = static_cast<float *>(ws);
  }
  dim3 grid((unsigned)tiles, (unsigned)S);
  const bool rsi = rs != nullptr && g_gemm_rsi && kch == 4;   // per-row-scaled segments applied in place (no running total)
  const bool folds = !rsi && (a.seg_stages > 0 || a.nsb > 0 || a.rs0 != nullptr);
  if (folds && a.res) { set_error("linear_c8: a residual cannot be combined with a folding (row-invariant / row-scaled) launch"); return MPN_EINVAL; }
  if (kch == 8) {
    if (folds) hipLaunchKernelGGL((gemm_c8_pf_kernel<8, true>), grid, dim3(256), (size_t)2 * 2 * 8 * 128 * 8 * 4, s, a);
    else hipLaunchKernelGGL((gemm_c8_pf_kernel<8, false>), grid, dim3(256), (size_t)2 * 2 * 8 * 128 * 8 * 4, s, a);
  } else {
    if (rsi) {
      int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_pf_kernel<4, false, true>), 2 * 2 * 4 * 128 * 8 * 4);
      if (rc_attr) return rc_attr;
      hipLaunchKernelGGL((gemm_c8_pf_kernel<4, false, true>), grid, dim3(256), (size_t)2 * 2 * 4 * 128 * 8 * 4, s, a);
    } else if (folds) hipLaunchKernelGGL((gemm_c8_pf_kernel<4, true>), grid, dim3(256), (size_t)2 * 2 * 4 * 128 * 8 * 4, s, a);
    else hipLaunchKernelGGL((gemm_c8_pf_kernel<4, false>), grid, dim3(256), (size_t)2 * 2 * 4 * 128 * 8 * 4, s, a);
  }
  MPN_CHECK_LAUNCH();
  if (!direct) {
    size_t total = (size_t)(a.NP / 8) * M;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, a.y, S, a.NP, a.Mp, M, N, d_bpk,
                       relu, d_y_c8, d_y_rm);
    MPN_CHECK_LAUNCH();
  }
  return MPN_OK;
}

// =================================================================================================
// fc6 on the bf16 matrix pipe with fp32 results: the three-plane split (VERDICT r5 task 2; models/vgg.lua:16,30)
// =================================================================================================
// An fp32 value is the exact sum of three bf16 values h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 8 + 8 + 8 significand
// bits), so x . w = sum over the nine plane products; the six with weight >= 2^-16 (hh, hm, mh, hl, lh, mm) are kept, the three at or below
// 2^-24 of the product (ml, lm, ll: the size of the fp32 rounding of the product itself) are dropped.  Each kept product is a
// v_mfma_f32_32x32x16_bf16 — bf16 x bf16 is exact in fp32, the accumulation is the MFMA's fp32 — and the bf16 pipe runs 16x the fp32 MFMA rate:
// six instructions of 32 cycles replace eight fp32 MFMAs of 64 (K = 16), 2.67x less matrix time for 1.5x the operand bytes.
//
// Kernel: block = 256 (N, weight rows) x 256 (M, ROI rows) x one k16 step per stage, four waves of 128 x 128 (16 accumulators = all 256 AGPRs, one
// wave per SIMD); per stage 48 KiB of operands (3 planes x 2 chunks x 256 rows x 16 B, both sides) arrive by LDS-DMA into a two-slot ring (stage st + 2 goes
// into stage st's slot as soon as stage st starts: its fragments are already in registers), the fragments of the next stage are read between
// the 96 MFMAs of the current one (one LDS or DMA instruction per MFMA slot, no VALU in the loop), ONE barrier per stage.  15.6 B / clock / CU of DMA, 31 B / clock / CU of LDS reads.
// Split-K over a FIXED number of K ranges (a function of K alone: a row's summation order does not depend on the rows it is batched with), partial
// slabs reduced in split order by splitk_reduce_kernel, which also adds the bias and applies the ReLU.
typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 s3_hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float s3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int s3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned s3_pack2(float lo, float hi) {  // bits 0-15 = bf16(lo), 16-31 = bf16(hi), round to nearest even (hardware conversion)
  const s3_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, s3_hwbf16x2));
}
__device__ __forceinline__ float s3_lo(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float s3_hi(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// fp32 C8 matrix [Kc][rows_src][8] -> three bf16 planes [3][Kc][rows_dst][8] (rows beyond rows_valid are left untouched: the planes are
// allocated zeroed).  One thread per 8-float record.
__global__ __launch_bounds__(256) void split3_planes_kernel(const float *__restrict__ src, int Kc, int Kc_valid, int rows_src, int rows_valid, int rows_dst,
                                                            unsigned short *__restrict__ dst) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)Kc_valid * rows_valid) return;
  const int r = (int)(t % rows_valid); const size_t kc = t / rows_valid;
  const f32x4 a = *reinterpret_cast<const f32x4 *>(src + (kc * rows_src + r) * 8), b = *reinterpret_cast<const f32x4 *>(src + (kc * rows_src + r) * 8 + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = v[2 * i], x1 = v[2 * i + 1];
    h[i] = s3_pack2(x0, x1);
    const float r0 = x0 - s3_lo(h[i]), r1 = x1 - s3_hi(h[i]);       // exact (Sterbenz-type cancellation: h is x rounded to 8 bits)
    m[i] = s3_pack2(r0, r1);
    l[i] = s3_pack2(r0 - s3_lo(m[i]), r1 - s3_hi(m[i]));
  }
  const size_t plane = (size_t)Kc * rows_dst * 8, rec = (kc * rows_dst + r) * 8;
  *reinterpret_cast<s3_u32x4 *>(dst + rec) = s3_u32x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<s3_u32x4 *>(dst + plane + rec) = s3_u32x4{m[0], m[1], m[2], m[3]};
  *reinterpret_cast<s3_u32x4 *>(dst + 2 * plane + rec) = s3_u32x4{l[0], l[1], l[2], l[3]};
}

struct Split3Args {
  const unsigned short *xp; int Mp;   // [3][Kc][Mp][8] bf16, Mp % 128 == 0
  const unsigned short *wp; int NP;   // [3][Kc][NP][8] bf16, NP % 256 == 0
  float *part; int part_np, part_mp;  // partial slabs [S][part_np / 8][part_mp][8] fp32 (the fp32 GEMM's slab geometry: splitk_reduce_kernel reads them)
  int Kc, steps, steps_per_split, n_mt, n_nt, M, N;
};

constexpr int S3_TN = 256, S3_TM = 128;                       // block tile: weight rows x ROI rows
constexpr int S3_A_BYTES = 3 * 2 * S3_TN * 16;                // 24 KiB: [plane][chunk][row] 16-byte records
constexpr int S3_B_BYTES = 3 * 2 * S3_TM * 16;                // 12 KiB
constexpr int S3_STAGE_BYTES = S3_A_BYTES + S3_B_BYTES;       // 36 KiB per k16 stage
constexpr int S3_RING = 3;                                    // 108 KiB: stage st + 2 lands while st and st + 1 are being read

// TWO accumulator sets.  acc (AGPRs) takes the h.h products — the sum to 16 bits of each factor —, acc2 (VGPRs) the five correction products, 2^-8 and
// 2^-16 of it.  Added into ONE running sum (the first form of this kernel), each of the six products of a k16 step is rounded at the ulp of the big
// sum: measured at full size against a float64 head, 3.4-3.7e-5 on logits of 16 where the fp32 MFMA pipeline has 2.0e-5, and one full-size parity
// test at 1.03e-4 of its 1e-4.  Kept apart, the corrections round at 2^-8 of that and the main sum takes one addition per k16 step.  acc + acc2
// in the epilogue.  Cost: one block per CU (one wave per SIMD) instead of two; the fragments are single-buffered to make room, each plane's
// registers reloaded for the next stage right after its last product of this one (order h.h, h.l, l.h, h.m, m.h, m.m).
__global__ __launch_bounds__(256, 1) void gemm_c8_split3_kernel(Split3Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wn = wave >> 1, wm = wave & 1;                          // wave tile: 128 weight rows x 64 ROI rows
  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);           // the blocks of one XCD walk neighbouring tiles
  const int nt = b / a.n_mt, mt = b - nt * a.n_mt;                 // all row tiles of one weight tile back to back: the weights are the big operand
  const int n0 = nt * S3_TN, m0 = mt * S3_TM;
  const int split = blockIdx.y;
  const int st0 = split * a.steps_per_split, st1 = min(a.steps, st0 + a.steps_per_split);
  if (st0 >= st1) return;

  // DMA: a stage is 24 + 12 one-KiB wave-loads (64 rows of one (operand, plane, chunk) each); wave w issues items w, w + 4, ...: 9 per stage
  const unsigned dma_lane = (unsigned)(lane * 16);
  const unsigned lds0 = lds_byte_addr(lds);
  const size_t w_plane = (size_t)a.Kc * a.NP * 8, x_plane = (size_t)a.Kc * a.Mp * 8;   // in bf16 elements
  const unsigned short *const w_tile = a.wp + (size_t)n0 * 8, *const x_tile = a.xp + (size_t)m0 * 8;
  auto issue = [&](int st, int slot, int j) {  // j = 0 .. 8: this wave's j-th item of the stage = item 4 j + wave (j compile-time: no branch)
    if (j < 6) {                               // weights: (plane, chunk) = j, row quarter = wave
      const int pl = j >> 1, ch = j & 1;
      const unsigned short *src = w_tile + pl * w_plane + (size_t)(2 * st + ch) * a.NP * 8 + (size_t)wave * 64 * 8;
      glds16_saddr(reinterpret_cast<const float *>(src), dma_lane, lds0 + (unsigned)(slot * S3_STAGE_BYTES + (j * S3_TN + wave * 64) * 16));
    } else {                                   // ROI rows: (plane, chunk) = 2 (j - 6) + wave / 2, row half = wave % 2
      const int pc = 2 * (j - 6) + (wave >> 1), q = wave & 1, pl = pc >> 1, ch = pc & 1;
      const unsigned short *src = x_tile + pl * x_plane + (size_t)(2 * st + ch) * a.Mp * 8 + (size_t)q * 64 * 8;
      glds16_saddr(reinterpret_cast<const float *>(src), dma_lane, lds0 + (unsigned)(slot * S3_STAGE_BYTES + S3_A_BYTES + (pc * S3_TM + q * 64) * 16));
    }
  };

  f32x16 acc[4][2], acc2[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[mi][ni][r] = 0.0f; acc2[mi][ni][r] = 0.0f; }

  // fragments: A (weights) fragment (plane, mi) = the record of row wn * 128 + mi * 32 + l31, chunk `half`; B (ROI rows) (plane, ni) = row
  // wm * 64 + ni * 32 + l31: 18 fragments = 72 VGPRs beside the 128 + 128 accumulator registers.
  typedef const s3_bf16x8 __attribute__((address_space(3))) *lds_frag_ptr;
  unsigned fbase[S3_RING][2];  // LDS byte addresses, opaque: every read is base + a 16-bit immediate (plane * 8 / 4 KiB + sub-tile * 512 B)
#pragma unroll
  for (int sl = 0; sl < S3_RING; ++sl) {
    fbase[sl][0] = lds0 + (unsigned)(sl * S3_STAGE_BYTES + (half * S3_TN + wn * 128 + l31) * 16);
    fbase[sl][1] = lds0 + (unsigned)(sl * S3_STAGE_BYTES + S3_A_BYTES + (half * S3_TM + wm * 64 + l31) * 16);
    asm volatile("" : "+v"(fbase[sl][0]));
    asm volatile("" : "+v"(fbase[sl][1]));
  }
  auto rda = [&](int slot, int pl, int i) -> s3_bf16x8 { return *(lds_frag_ptr)(size_t)(fbase[slot][0] + (unsigned)(pl * (2 * S3_TN * 16) + i * (32 * 16))); };
  auto rdb = [&](int slot, int pl, int i) -> s3_bf16x8 { return *(lds_frag_ptr)(size_t)(fbase[slot][1] + (unsigned)(pl * (2 * S3_TM * 16) + i * (32 * 16))); };
  s3_bf16x8 ha[4], hb[2], ma[4], mb[2], la[4], lb[2];

  // prologue: stages st0 (slot 0) and st0 + 1 (slot 1) in flight, the fragments of st0 in registers
#pragma unroll
  for (int j = 0; j < 9; ++j) issue(st0, 0, j);
#pragma unroll
  for (int j = 0; j < 9; ++j) issue(min(st0 + 1, st1 - 1), 1, j);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { ha[i] = rda(0, 0, i); ma[i] = rda(0, 1, i); la[i] = rda(0, 2, i); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { hb[i] = rdb(0, 0, i); mb[i] = rdb(0, 1, i); lb[i] = rdb(0, 2, i); }

  // One stage = six products of 8 MFMAs: h.h -> acc; h.l, l.h, h.m, m.h, m.m -> acc2.  Between the MFMAs one LDS / DMA instruction per slot:
  // the 9 DMA items of stage st + 2 (ring slot S2: read out two stages ago) during h.h / h.l; the next stage's fragments (slot S1: its DMA was
  // waited for and published by the barrier at the top) into each plane's registers right after the plane's last product: lb after h.l, la after
  // l.h, ha after h.m, hb after m.h, ma / mb during the next stage's h.h (they are first needed by its fourth product).
  // BRANCH-FREE: past the last stage the prefetches repeat the last stage (clamped index; nobody reads them).
  const int st_last = st1 - 1;
  auto body = [&](int st, auto slot_tag, bool first) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr int S1 = (SLOT + 1) % S3_RING, S2 = (SLOT + 2) % S3_RING;
    const int st2 = min(st + 2, st_last);
    dma_wait_all();        // stage st + 1 has landed (this wave's pieces; issued a whole stage ago) ...
    __syncthreads();       // ... and everyone's; every wave has also finished stage st - 1, whose fragments were the last reads of slot S2's old tenant
#define S3_MFMA_INTO(ACC, A, B, MI, NI)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    ACC[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, ACC[MI][NI], 0, 0, 0);                   \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // h.h -> the main sum   (+ this stage's m fragments, which the previous stage's m.m was still using; not in the first stage: the prologue loaded them)
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc, ha[mi], hb[ni], mi, ni)
      if (!first) { if (t < 4) ma[t] = rda(SLOT, 1, t); else if (t < 6) mb[t - 4] = rdb(SLOT, 1, t - 4); }
      if (t >= 6) issue(st2, S2, t - 6);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // h.l
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc2, ha[mi], lb[ni], mi, ni)
      if (t < 7) issue(st2, S2, 2 + t);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // l.h
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc2, la[mi], hb[ni], mi, ni)
      if (t < 2) lb[t] = rdb(S1, 2, t);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // h.m
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc2, ha[mi], mb[ni], mi, ni)
      if (t < 4) la[t] = rda(S1, 2, t);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // m.h
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc2, ma[mi], hb[ni], mi, ni)
      if (t < 4) ha[t] = rda(S1, 0, t);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {  // m.m
      const int mi = t >> 1, ni = t & 1;
      S3_MFMA_INTO(acc2, ma[mi], mb[ni], mi, ni)
      if (t < 2) hb[t] = rdb(S1, 0, t);
    }
#undef S3_MFMA_INTO
  };
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;
  int st = st0;
  body(st, T0{}, true);
  ++st;
  while (st < st1) {
    body(st, T1{}, false); if (++st >= st1) break;
    body(st, T2{}, false); if (++st >= st1) break;
    body(st, T0{}, false); ++st;
  }
  dma_wait_all();  // the clamped prefetches of the last two stages still write this block's LDS: they must have landed before the block ends

  // epilogue: the partial slab of this split, in the fp32 GEMM's slab geometry
  float *yb = a.part + (size_t)split * (a.part_np / 8) * a.part_mp * 8;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 128 + mi * 32 + g * 8;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int m = m0 + wm * 64 + ni * 32 + l31;
        const f32x4 v = {acc[mi][ni][g * 4 + 0] + acc2[mi][ni][g * 4 + 0], acc[mi][ni][g * 4 + 1] + acc2[mi][ni][g * 4 + 1],
                         acc[mi][ni][g * 4 + 2] + acc2[mi][ni][g * 4 + 2], acc[mi][ni][g * 4 + 3] + acc2[mi][ni][g * 4 + 3]};
        if (n < a.part_np && m < a.part_mp) *reinterpret_cast<f32x4 *>(yb + ((size_t)(n / 8) * a.part_mp + m) * 8 + half * 4) = v;
      }
    }
}

size_t split3_plane_elems(int K, int rows) { return (size_t)3 * (round_up(K, 64) / 8) * round_up(rows, 256) * 8; }  // bf16 elements of a three-plane operand (rows padded to the 256-row weight tile)

// fp32 C8 matrix (K64 / 8 chunks of `rows_src` rows; the packed weights [K/8][NP][8] or the activations [K/8][Mp][8]) -> its three bf16 planes
int split3_planes(const float *d_c8, int K, int rows_src, int rows_valid, unsigned short *d_planes, hipStream_t s) {
  MPN_CHECK_ARG(d_c8 && d_planes && K > 0 && rows_valid > 0 && rows_valid <= rows_src);
  const int Kc = round_up(K, 64) / 8, Kc_valid = (K + 7) / 8;   // the chunks past K (up to the k16-step granularity) stay zero: the planes are allocated zeroed
  const size_t total = (size_t)Kc_valid * rows_valid;
  hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_c8, Kc, Kc_valid, rows_src, rows_valid, round_up(rows_src, 256), d_planes);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// y = relu?(x . w^T + b) with both operands given as three-plane splits; y in the fp32 C8 layout [NP / 8][Mp][8] (Mp = lin_mp(M))
int linear_c8_split3(const unsigned short *d_x3, int M, int K, const unsigned short *d_w3, const float *d_bpk, int N, int relu, float *d_y_c8, hipStream_t s) {
  MPN_CHECK_ARG(d_x3 && d_w3 && d_bpk && d_y_c8 && M > 0 && K > 0 && N > 0);
  Split3Args a{};
  const int Mp = lin_mp(M), NP = lin_np(N);
  a.xp = d_x3; a.Mp = round_up(Mp, 256); a.wp = d_w3; a.NP = round_up(NP, 256);   // (the planes' row pitch: split3_plane_elems)
  a.part_np = NP; a.part_mp = Mp;
  a.Kc = round_up(K, 64) / 8; a.steps = a.Kc / 2;
  a.n_mt = Mp / S3_TM; a.n_nt = a.NP / S3_TN; a.M = M; a.N = N;
  // K ranges from (K, N) alone — never from M: a row's summation order must not depend on the rows it is batched with.  As many as make 512 blocks
  // (two per CU) at the usual 8 row tiles (1000 ROIs), at most 8, each >= 32 k16 steps: fc6 (16 weight tiles, 1568 steps) 4 ranges of 392, fc7 (256 steps) 4 of 64
  int S = 512 / (a.n_nt * 8);
  if (g_split3_ranges > 0) S = g_split3_ranges;   // (mpn_debug_set_split3_ranges: accuracy / timing experiments)
  if (S > 8) S = 8;
  if (S > a.steps / 32) S = a.steps / 32;
  if (S < 1) S = 1;
  a.steps_per_split = cdiv(a.steps, S);
  S = cdiv(a.steps, a.steps_per_split);
  const size_t need = (size_t)S * (NP / 8) * Mp * 8 * sizeof(float);
  void *ws = nullptr;
  { int rc_ws = scratch_get(t_gemm_splitk_slot, need, s, &ws); if (rc_ws) return rc_ws; }
  a.part = static_cast<float *>(ws);
  { int rc_attr = set_max_dyn_lds(reinterpret_cast<const void *>(gemm_c8_split3_kernel), S3_RING * S3_STAGE_BYTES); if (rc_attr) return rc_attr; }
  hipLaunchKernelGGL(gemm_c8_split3_kernel, dim3((unsigned)(a.n_mt * a.n_nt), (unsigned)S), dim3(256), (size_t)S3_RING * S3_STAGE_BYTES, s, a);
  MPN_CHECK_LAUNCH();
  const size_t total = (size_t)(NP / 8) * M;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, a.part, S, NP, Mp, M, N, d_bpk, relu, d_y_c8, (float *)nullptr);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// =================================================================================================
// packers / converters / pooling
// =================================================================================================
__global__ void pack_conv_w_kernel(const float *__restrict__ w, const float *__restrict__ b, int Cin, int Cout, int CoutP,
                                   int nchunks, float *__restrict__ wpk, float *__restrict__ bpk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nchunks * 9 * CoutP * 8;
  if (t < (size_t)CoutP) bpk[t] = (t < (size_t)Cout && b) ? b[t] : 0.0f;
  if (t >= total) return;
  int j = (int)(t & 7);
  size_t r = t >> 3;
  int co = (int)(r % CoutP); r /= CoutP;
  int tap = (int)(r % 9);
  int ch = (int)(r / 9);
  int ci = ch * 8 + j;
  wpk[t] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 9 + tap] : 0.0f;
}

// Winograd filter transform U = G g G^T (computed in double, rounded once) into [Cin8/8][16][CoutP][8]
__global__ void pack_conv_w_wino_kernel(const float *__restrict__ w, int Cin, int Cout, int CoutP, int nchunks, float *__restrict__ wino) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nchunks * 16 * CoutP * 8;
  if (t >= total) return;
  int j = (int)(t & 7);
  size_t r = t >> 3;
  int co = (int)(r % CoutP); r /= CoutP;
  int comp = (int)(r % 16);
  int ch = (int)(r / 16);
  // LDS bank swizzle of the kernel's operand records (see conv3x3_wino_kernel): the record of cout row r stores its two
  // 4-channel halves swapped when bit 3 of r is set, so that the 16 lanes of a ds_read_b128 lane group hit 16 distinct slots
  int ci = ch * 8 + (j ^ (((co >> 3) & 1) << 2));
  float v = 0.0f;
  if (co < Cout && ci < Cin) {
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const float *g = w + ((size_t)co * Cin + ci) * 9;
    const int xi = comp >> 2, nu = comp & 3;
    double acc = 0.0;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) acc += G[xi][ky] * (double)g[ky * 3 + kx] * G[nu][kx];
    v = (float)acc;
  }
  wino[t] = v;
}

int pack_conv_weights_wino(const float *d_w, int Cin, int Cout, float *d_wino, hipStream_t s) {
  MPN_CHECK_ARG(d_w && d_wino && Cin > 0 && Cout > 0);
  const int nchunks = (Cin + 7) / 8, CoutP = conv_coutp(Cout);
  size_t total = (size_t)nchunks * 16 * CoutP * 8;
  hipLaunchKernelGGL(pack_conv_w_wino_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_w, Cin, Cout, CoutP, nchunks, d_wino);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

int pack_conv_weights(const float *d_w, const float *d_b, int Cin, int Cout, float *d_wpk, float *d_bpk, hipStream_t s) {
  MPN_CHECK_ARG(d_w && d_wpk && d_bpk && Cin > 0 && Cout > 0);
  int nch = (Cin + 7) / 8, CoutP = conv_coutp(Cout);
  size_t total = (size_t)nch * 9 * CoutP * 8;
  hipLaunchKernelGGL(pack_conv_w_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_w, d_b, Cin, Cout, CoutP, nch, d_wpk, d_bpk);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void pack_lin_w_kernel(const float *__restrict__ w, const float *__restrict__ b, int K, int N, int NP, int nq,
                                  int inner, float *__restrict__ wpk, float *__restrict__ bpk) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nq * NP * 8;
  if (t < (size_t)NP) bpk[t] = (t < (size_t)N && b) ? b[t] : 0.0f;
  if (t >= total) return;
  int j = (int)(t & 7);
  size_t r = t >> 3;
  int n = (int)(r % NP);
  int q = (int)(r / NP);
  long k = ((long)(q / inner) * 8 + j) * inner + (q % inner);
  wpk[t] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.0f;
}

int pack_linear_weights(const float *d_w, const float *d_b, int K, int N, int inner, float *d_wpk, float *d_bpk, hipStream_t s) {
  MPN_CHECK_ARG(d_w && d_wpk && d_bpk && K > 0 && N > 0 && inner > 0);
  int K32 = round_up(K, 64), nq = K32 / 8, NP = lin_np(N);
  MPN_CHECK_ARG(inner == 1 || (K % (8 * inner)) == 0);
  size_t total = (size_t)nq * NP * 8;
  hipLaunchKernelGGL(pack_lin_w_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_w, d_b, K, N, NP, nq, inner, d_wpk, d_bpk);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void nchw_to_c8p_kernel(const float *__restrict__ in, int C, int H, int W, float *__restrict__ out, int Hp, int Wp) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)((C + 7) / 8) * H * W;
  if (t >= total) return;
  int x = (int)(t % W); size_t r = t / W;
  int y = (int)(r % H); int cb = (int)(r / H);
  f32x4 lo, hi;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c0 = cb * 8 + j, c1 = c0 + 4;
    lo[j] = c0 < C ? in[((size_t)c0 * H + y) * W + x] : 0.0f;
    hi[j] = c1 < C ? in[((size_t)c1 * H + y) * W + x] : 0.0f;
  }
  float *o = out + (((size_t)cb * Hp + y + 1) * Wp + x + 1) * 8;
  *reinterpret_cast<f32x4 *>(o) = lo;
  *reinterpret_cast<f32x4 *>(o + 4) = hi;
}

int nchw_to_c8p(const float *d_in, int C, int H, int W, Act out, hipStream_t s) {
  MPN_CHECK_ARG(d_in && out.p && out.C == C && out.H == H && out.W == W);
  size_t total = (size_t)out.Cb() * H * W;
  hipLaunchKernelGGL(nchw_to_c8p_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_in, C, H, W, out.p, out.Hp, out.Wp);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void c8p_to_nchw_kernel(const float *__restrict__ in, int C, int H, int W, int Hp, int Wp, float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)C * H * W;
  if (t >= total) return;
  int x = (int)(t % W); size_t r = t / W;
  int y = (int)(r % H); int c = (int)(r / H);
  out[t] = in[(((size_t)(c >> 3) * Hp + y + 1) * Wp + x + 1) * 8 + (c & 7)];
}

int c8p_to_nchw(Act in, float *d_out, hipStream_t s) {
  MPN_CHECK_ARG(in.p && d_out);
  size_t total = (size_t)in.C * in.H * in.W;
  hipLaunchKernelGGL(c8p_to_nchw_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, in.p, in.C, in.H, in.W, in.Hp, in.Wp, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void rowmajor_to_c8_kernel(const float *__restrict__ x, int M, int K, int Mp, float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [K8/8][M] records
  size_t total = (size_t)((K + 7) / 8) * M;
  if (t >= total) return;
  int m = (int)(t % M); int q = (int)(t / M);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { int k = q * 8 + j; v[j] = k < K ? x[(size_t)m * K + k] : 0.0f; }
  float *o = out + ((size_t)q * Mp + m) * 8;
  *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4 *>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

int rowmajor_to_c8(const float *d_x, int M, int K, float *d_c8, hipStream_t s) {
  MPN_CHECK_ARG(d_x && d_c8 && M > 0 && K > 0);
  size_t total = (size_t)((K + 7) / 8) * M;
  hipLaunchKernelGGL(rowmajor_to_c8_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_x, M, K, lin_mp(M), d_c8);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void c8_to_rowmajor_kernel(const float *__restrict__ c8, int M, int N, int Mp, float *__restrict__ y) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)M * N;
  if (t >= total) return;
  int n = (int)(t % N); int m = (int)(t / N);
  y[t] = c8[((size_t)(n >> 3) * Mp + m) * 8 + (n & 7)];
}

int c8_to_rowmajor(const float *d_c8, int M, int N, float *d_y, hipStream_t s) {
  MPN_CHECK_ARG(d_c8 && d_y && M > 0 && N > 0);
  size_t total = (size_t)M * N;
  hipLaunchKernelGGL(c8_to_rowmajor_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_c8, M, N, lin_mp(M), d_y);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// ImageTransformer fused with the C8P conversion (3 real channels + 5 zero channels)
__global__ void image_transform_c8p_kernel(const float *__restrict__ in, int H, int W, int s0, int s1, int s2, double scale,
                                           double m0, double m1, double m2, double d0, double d1, double d2, int has_std,
                                           float *__restrict__ out, int Hp, int Wp) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t plane = (size_t)H * W;
  if (t >= plane) return;
  int x = (int)(t % W), y = (int)(t / W);
  double v0 = (double)in[(size_t)s0 * plane + t], v1 = (double)in[(size_t)s1 * plane + t], v2 = (double)in[(size_t)s2 * plane + t];
  if (scale != 1.0) { v0 = v0 * scale; v1 = v1 * scale; v2 = v2 * scale; }
  v0 = v0 + (-m0); v1 = v1 + (-m1); v2 = v2 + (-m2);
  if (has_std) { v0 = v0 / d0; v1 = v1 / d1; v2 = v2 / d2; }
  float *o = out + (((size_t)y + 1) * Wp + x + 1) * 8;
  *reinterpret_cast<f32x4 *>(o) = f32x4{(float)v0, (float)v1, (float)v2, 0.0f};
  *reinterpret_cast<f32x4 *>(o + 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

int image_transform_c8p(const float *d_in, int H, int W, const int *swap, double scale, const double *mean, const double *std,
                        int has_std, Act out, hipStream_t s) {
  MPN_CHECK_ARG(d_in && out.p && out.H == H && out.W == W && out.C <= 8);
  size_t plane = (size_t)H * W;
  hipLaunchKernelGGL(image_transform_c8p_kernel, dim3((unsigned)cdiv_sz(plane, 256)), dim3(256), 0, s, d_in, H, W, swap[0], swap[1],
                     swap[2], scale, mean[0], mean[1], mean[2], has_std ? std[0] : 1.0, has_std ? std[1] : 1.0,
                     has_std ? std[2] : 1.0, has_std, out.p, out.Hp, out.Wp);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// Re-lays the zero halo of up to kHaloMax C8P activations for a new image size in ONE launch: every record of a plane that is not one of the
// H x W interior pixels (row 0, rows H + 1 .. Hp - 1, column 0, columns W + 1 .. Wp - 1).  The interior is rewritten by the producing layer
// before anything reads it; the halo is what the 3x3 padding and the ragged tile edges read.  (Round 5: this replaces one hipMemsetAsync of
// the WHOLE allocation per activation — 19 runtime blits, ~1 GB at a 1000 x 1000 cap — which cost a mixed-size stream 0.2 ms per size change
// stand-alone and 1.9 ms inside bench.py, where the blits' cross-queue synchronisation went through more hardware queues.)
__global__ void c8p_zero_halos_kernel(HaloTable t) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= t.total) return;
  int k = 0;
#pragma unroll 1
  while (k + 1 < t.n && g >= t.first[k + 1]) ++k;
  const HaloDesc d = t.d[k];
  size_t r = g - t.first[k];          // half-record index inside activation k
  const int half = (int)(r & 1); r >>= 1;
  const size_t per_plane = (size_t)d.Wp * (d.Hp - d.H) + (size_t)d.H * (d.Wp - d.W);
  const int cb = (int)(r / per_plane);
  size_t h = r - (size_t)cb * per_plane;
  int row, col;
  const size_t full_rows = (size_t)d.Wp * (d.Hp - d.H);   // row 0 and rows H + 1 .. Hp - 1, whole
  if (h < full_rows) {
    const int ri = (int)(h / d.Wp);
    col = (int)(h - (size_t)ri * d.Wp);
    row = ri == 0 ? 0 : d.H + ri;
  } else {
    h -= full_rows;
    const int wd = d.Wp - d.W;                           // column 0 and columns W + 1 .. Wp - 1 of rows 1 .. H
    const int ri = (int)(h / wd), ci = (int)(h - (size_t)ri * wd);
    row = 1 + ri;
    col = ci == 0 ? 0 : d.W + ci;
  }
  *reinterpret_cast<f32x4 *>(d.p + (((size_t)cb * d.Hp + row) * d.Wp + col) * 8 + half * 4) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

int c8p_zero_halos(const Act *acts, int n, hipStream_t s) {
  MPN_CHECK_ARG(acts && n >= 0);
  for (int i0 = 0; i0 < n; i0 += kHaloMax) {
    HaloTable t{};
    t.n = n - i0 < kHaloMax ? n - i0 : kHaloMax;
    size_t tot = 0;
    for (int k = 0; k < t.n; ++k) {
      const Act &a = acts[i0 + k];
      MPN_CHECK_ARG(a.p && a.H > 0 && a.W > 0 && a.Hp > a.H && a.Wp > a.W);
      t.d[k] = HaloDesc{a.p, a.H, a.W, a.Hp, a.Wp};
      t.first[k] = tot;
      tot += ((size_t)a.Wp * (a.Hp - a.H) + (size_t)a.H * (a.Wp - a.W)) * a.Cb() * 2;
    }
    t.total = tot;
    if (!tot) continue;
    hipLaunchKernelGGL(c8p_zero_halos_kernel, dim3((unsigned)cdiv_sz(tot, 256)), dim3(256), 0, s, t);
    MPN_CHECK_LAUNCH();
  }
  return MPN_OK;
}

__global__ void maxpool2x2_c8p_kernel(const float *__restrict__ in, int H, int W, int Hp, int Wp, int Cb, float *__restrict__ out,
                                      int Ho, int Wo, int Hpo, int Wpo) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * Ho * Wo * 2;  // half-records (float4)
  if (t >= total) return;
  int h = (int)(t & 1); size_t r = t >> 1;
  int x = (int)(r % Wo); r /= Wo;
  int y = (int)(r % Ho); int cb = (int)(r / Ho);
  f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy < H && xx < W) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((size_t)cb * Hp + yy + 1) * Wp + xx + 1) * 8 + h * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
  *reinterpret_cast<f32x4 *>(out + (((size_t)cb * Hpo + y + 1) * Wpo + x + 1) * 8 + h * 4) = m;
}

int maxpool2x2_c8p(Act in, Act out, hipStream_t s) {
  MPN_CHECK_ARG(in.p && out.p && out.C == in.C && out.H == (in.H + 1) / 2 && out.W == (in.W + 1) / 2);
  size_t total = (size_t)in.Cb() * out.H * out.W * 2;
  hipLaunchKernelGGL(maxpool2x2_c8p_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, in.p, in.H, in.W, in.Hp, in.Wp, in.Cb(),
                     out.p, out.H, out.W, out.Hp, out.Wp);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

__global__ void maxpool2x2_nchw_kernel(const float *__restrict__ in, size_t BC, int H, int W, float *__restrict__ out) {
  int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = BC * Ho * Wo;
  if (t >= total) return;
  int x = (int)(t % Wo); size_t r = t / Wo;
  int y = (int)(r % Ho); size_t c = r / Ho;
  float m = -INFINITY;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      int yy = 2 * y + dy, xx = 2 * x + dx;
      if (yy < H && xx < W) { float v = in[(c * H + yy) * W + xx]; m = v > m ? v : m; }
    }
  out[t] = m;
}

// ROI max-pool: C8P feature map -> C8 matrix [cb*PH*PW + bin][Mp][8].  One thread per
// (cb, bin, roi) half-record; roi fastest so a wave writes 1 KiB contiguous.
__global__ __launch_bounds__(256) void roi_pool_c8_kernel(const float *__restrict__ feat, int C, int H, int W, int Hp, int Wp,
                                                          const float *__restrict__ rois, int roi_stride, int N, int PH, int PW,
                                                          float scale, RoiRule rr, float *__restrict__ xc8,
                                                          int Mp, int32_t *__restrict__ argmax) {
  const int Cb = (C + 7) / 8, PP = PH * PW;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * PP * N * 2;
  if (t >= total) return;
  int h = (int)(t & 1); size_t r = t >> 1;
  int n = (int)(r % N); r /= N;
  int bin = (int)(r % PP); int cb = (int)(r / PP);
  int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, rr, H, W, PH, PW, ph, pw, hs, he, ws, we);
  bool empty = (he <= hs) || (we <= ws);
  f32x4 m = empty ? f32x4{0, 0, 0, 0} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int mi[4] = {-1, -1, -1, -1};
  const float *fp = feat + (size_t)cb * Hp * Wp * 8 + h * 4;
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(fp + ((size_t)(y + 1) * Wp + x + 1) * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > m[e]) { m[e] = v[e]; mi[e] = y * W + x; }
    }
  *reinterpret_cast<f32x4 *>(xc8 + (((size_t)cb * PP + bin) * Mp + n) * 8 + h * 4) = m;
  if (argmax) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int c = cb * 8 + h * 4 + e;
      if (c < C) argmax[(((size_t)n * C + c) * PH + ph) * PW + pw] = mi[e];
    }
  }
}

// nn.Normalize(2) + MulConstant over one pooled map of the skip concat (model_utils.lua:216-223,236-241):
// per ROI, (x / sqrt(sum x^2 + 1e-10)) * mul over all C*PH*PW values of the map, which occupies `nrec` 8-float
// records (pitch Mp) of a C8 matrix.  Three HBM-streaming passes, all coalesced over the ROI index and
// deterministic: (1) per (record group, roi) partial sums, (2) per roi: add the partials in group order -> norm,
// (3) elementwise scale.
__global__ __launch_bounds__(256) void l2norm_partial_kernel(const float *__restrict__ x, int nrec, int per, int Mp, int N,
                                                             float *__restrict__ part) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
  if (n >= N) return;
  const int r0 = g * per, r1 = min(nrec, r0 + per);
  float ss = 0.0f;
  for (int r = r0; r < r1; ++r) {
    const float *q = x + ((size_t)r * Mp + n) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(q), b = *reinterpret_cast<const f32x4 *>(q + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) ss += a[e] * a[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) ss += b[e] * b[e];
  }
  part[(size_t)g * N + n] = ss;
}
__global__ void l2norm_finish_kernel(const float *__restrict__ part, int G, int N, float *__restrict__ nrm) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float ss = 0.0f;
  for (int g = 0; g < G; ++g) ss += part[(size_t)g * N + n];
  nrm[n] = sqrtf(ss + 1e-10f);
}
__global__ __launch_bounds__(256) void l2norm_apply_kernel(float *__restrict__ x, size_t nrec, int Mp, int N,
                                                           const float *__restrict__ nrm, float mul) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (record, roi, half)
  const size_t total = nrec * (size_t)N * 2;
  if (t >= total) return;
  const int h = (int)(t & 1);
  const size_t q = t >> 1;
  const int n = (int)(q % N);
  const size_t r = q / N;
  float *p = x + (r * Mp + n) * 8 + h * 4;
  f32x4 v = *reinterpret_cast<f32x4 *>(p);
  if (nrm) {
    const float d = nrm[n];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] / d) * mul;
  } else {  // nn.MulConstant(mul): the isNormalized = false branch of conv345Combine (model_utils.lua:222-223)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * mul;
  }
  *reinterpret_cast<f32x4 *>(p) = v;
}


int l2norm_scale_c8(float *d_x_c8, int n_records, int Mp, int N, float mul, hipStream_t s) {
  MPN_CHECK_ARG(d_x_c8 && n_records > 0 && N > 0 && Mp >= N);
  const int per = 49;  // one channel block's bins per partial sum
  const int G = cdiv(n_records, per);
  const size_t need = ((size_t)G + 1) * N * sizeof(float);
  void *ws = nullptr;
  { int rc_ws = scratch_get(SCR_L2NORM, need, s, &ws); if (rc_ws) return rc_ws; }
  float *part = static_cast<float *>(ws), *nrm = part + (size_t)G * N;
  hipLaunchKernelGGL(l2norm_partial_kernel, dim3(cdiv(N, 256), G), dim3(256), 0, s, d_x_c8, n_records, per, Mp, N, part);
  MPN_CHECK_LAUNCH();
  hipLaunchKernelGGL(l2norm_finish_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, part, G, N, nrm);
  MPN_CHECK_LAUNCH();
  const size_t total = (size_t)n_records * N * 2;
  hipLaunchKernelGGL(l2norm_apply_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_x_c8, (size_t)n_records, Mp, N, nrm, mul);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

int mul_const_c8(float *d_x_c8, int n_records, int Mp, int N, float mul, hipStream_t s) {
  MPN_CHECK_ARG(d_x_c8 && n_records > 0 && N > 0 && Mp >= N);
  if (mul == 1.0f) return MPN_OK;  // x * 1.0f == x bit for bit
  const size_t total = (size_t)n_records * N * 2;
  hipLaunchKernelGGL(l2norm_apply_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_x_c8, (size_t)n_records, Mp, N, (const float *)nullptr, mul);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// ---- ROI max-pool through vertical range-max tables (MultiPathNet head) ----------------------------------------
// The skip-pool towers pool the stride-4 / stride-8 maps over Foveal regions up to 4x the ROI: a bin can cover hundreds
// of pixels (a 21 x 36 window on conv3 for a full-image x4 region), and the direct kernel visits every one of them for
// each of 11 (tower, map) pools.  Level k of the table holds T_k[y][x] = max(feat[y .. y + 2^k - 1][x]) (rows clipped at
// H), built once per image by doubling; a bin [hs, he) x [ws, we) is then max over x of max(T_k[hs][x], T_k[he - 2^k][x])
// with k = floor(log2(he - hs)): 2 (we - ws) reads instead of (he - hs)(we - ws).  max is exact, so the result is
// identical to the direct kernel's (no argmax: the inference pipeline does not need it).
__global__ void vmax_level_kernel(const float *__restrict__ prev, float *__restrict__ out, int Cb, int H, int W, int Hp, int Wp, int step) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * H * W * 2;
  if (t >= total) return;
  const int h = (int)(t & 1); size_t r = t >> 1;
  const int x = (int)(r % W); r /= W;
  const int y = (int)(r % H); const int cb = (int)(r / H);
  const size_t base = (size_t)cb * Hp * Wp * 8 + h * 4;
  const size_t o = base + ((size_t)(y + 1) * Wp + x + 1) * 8;
  f32x4 a = *reinterpret_cast<const f32x4 *>(prev + o);
  if (y + step < H) {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(prev + base + ((size_t)(y + step + 1) * Wp + x + 1) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = b[e] > a[e] ? b[e] : a[e];
  }
  *reinterpret_cast<f32x4 *>(out + o) = a;
}

int vmax_levels_for(int H) {  // levels 1 .. L with 2^L <= H
  int L = 0;
  while ((2 << L) <= H) ++L;
  return L;
}

int build_vmax_tables(Act feat, float *d_tables, hipStream_t s) {
  MPN_CHECK_ARG(feat.p && d_tables);
  const int L = vmax_levels_for(feat.H);
  const size_t total = (size_t)feat.Cb() * feat.H * feat.W * 2;
  const float *prev = feat.p;
  for (int k = 1; k <= L; ++k) {
    float *out = d_tables + (size_t)(k - 1) * feat.elems();
    hipLaunchKernelGGL(vmax_level_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, prev, out, feat.Cb(), feat.H, feat.W, feat.Hp,
                       feat.Wp, 1 << (k - 1));
    MPN_CHECK_LAUNCH();
    prev = out;
  }
  return MPN_OK;
}

__global__ __launch_bounds__(256) void roi_pool_c8_rmq_kernel(const float *__restrict__ feat, const float *__restrict__ tables, size_t level_elems,
                                                              int C, int H, int W, int Hp, int Wp, const float *__restrict__ rois,
                                                              int roi_stride, int N, int PH, int PW, float scale, RoiRule rr, float *__restrict__ xc8, int Mp) {
  const int Cb = (C + 7) / 8, PP = PH * PW;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * PP * N * 2;
  if (t >= total) return;
  int h = (int)(t & 1); size_t r = t >> 1;
  int n = (int)(r % N); r /= N;
  int bin = (int)(r % PP); int cb = (int)(r / PP);
  int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + (size_t)roi_stride * n;
  int hs, he, ws, we;
  roi_bin_bounds(ro, scale, rr, H, W, PH, PW, ph, pw, hs, he, ws, we);
  const bool empty = (he <= hs) || (we <= ws);
  f32x4 m = empty ? f32x4{0, 0, 0, 0} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  if (!empty) {
    const int k = 31 - __clz(he - hs);  // 2^k <= he - hs < 2^(k+1)
    const float *lvl = (k == 0) ? feat : tables + (size_t)(k - 1) * level_elems;
    const float *r0 = lvl + (size_t)cb * Hp * Wp * 8 + h * 4 + ((size_t)(hs + 1) * Wp + 1) * 8;
    const float *r1 = lvl + (size_t)cb * Hp * Wp * 8 + h * 4 + ((size_t)(he - (1 << k) + 1) * Wp + 1) * 8;
    for (int xb = ws; xb < we; xb += 4) {  // 8 independent loads in flight; positions past the window re-read its last column
      f32x4 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = min(xb + j, we - 1);
        a[j] = *reinterpret_cast<const f32x4 *>(r0 + (size_t)x * 8);
        b[j] = *reinterpret_cast<const f32x4 *>(r1 + (size_t)x * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = b[j][e] > a[j][e] ? b[j][e] : a[j][e];
          if (v > m[e]) m[e] = v;
        }
    }
  }
  *reinterpret_cast<f32x4 *>(xc8 + (((size_t)cb * PP + bin) * Mp + n) * 8 + h * 4) = m;
}

int roi_pool_c8_rmq(Act feat, const float *d_tables, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr, float *d_x_c8, hipStream_t s, int roi_stride, int Mp) {
  MPN_CHECK_ARG(feat.p && d_tables && d_rois && d_x_c8 && N > 0 && PH > 0 && PW > 0);
  size_t total = (size_t)feat.Cb() * PH * PW * N * 2;
  hipLaunchKernelGGL(roi_pool_c8_rmq_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, feat.p, d_tables, feat.elems(), feat.C, feat.H,
                     feat.W, feat.Hp, feat.Wp, d_rois, roi_stride, N, PH, PW, scale, rr, d_x_c8, Mp > 0 ? Mp : lin_mp(N));
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// ---- ROI max-pool from a pixel-major copy of the map --------------------------------------------------------------------------
// roi_pool_c8_kernel's wave reads 32 different ROIs' windows per load instruction: 32 cache lines, a quarter of each used — the vector
// L1's line rate, not HBM, is its limit (1.0 TB/s of algorithmic traffic).  With the map copied once per image to pixel-major order
// [y][x][C] (4.9 MB for conv5), a wave is ONE (roi, bin) and its 64 lanes are 256 consecutive channels: a load instruction is one
// contiguous 1 KiB (8 fully used lines), the window bounds are wave-uniform (no divergence), and the four waves of a block are four
// consecutive ROIs whose outputs are staged in LDS and leave as whole 128-byte lines of the fc6 operand.  Max is exact: the output
// is bit-identical to roi_pool_c8_kernel's.
__global__ void c8p_to_pixel_major_kernel(const float *__restrict__ in, int Cb, int H, int W, int Hp, int Wp, float *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)H * W * Cb * 2;
  if (t >= total) return;
  const int q = (int)(t % (Cb * 2)); const size_t px = t / (Cb * 2);   // q = cb * 2 + half: consecutive lanes write consecutive 16 bytes
  const int x = (int)(px % W), y = (int)(px / W);
  const int cb = q >> 1, h = q & 1;
  *reinterpret_cast<f32x4 *>(out + px * (size_t)Cb * 8 + q * 4) =
      *reinterpret_cast<const f32x4 *>(in + (size_t)cb * Hp * Wp * 8 + ((size_t)(y + 1) * Wp + x + 1) * 8 + h * 4);
}

template <int ABL>  // ABL: timing experiments (debug flavour): 1 no feature loads, 2 no stores, 4 no ROI decode (fixed 3x3 bin)
__global__ __launch_bounds__(256) void roi_pool_pm_kernel(const float *__restrict__ pm, int Cb, int H, int W, const float *__restrict__ rois,
                                                          int roi_stride, int N, int PH, int PW, float scale, RoiRule rr, float *__restrict__ xc8, int Mp) {
  __shared__ f32x4 stage[4][64];   // [roi of the quad][lane] = 4 channels
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 4, n = n0 + wave;
  const int bin = blockIdx.y, cq = blockIdx.z;   // cq: which 256-channel slice
  const int ph = bin / PW, pw = bin - ph * PW;
  const int C = Cb * 8;
  f32x4 m = f32x4{0, 0, 0, 0};
  if (n < N) {
    const float *ro = rois + (size_t)roi_stride * n;
    int hs, he, ws, we;
    roi_bin_bounds(ro, scale, rr, H, W, PH, PW, ph, pw, hs, he, ws, we);
    hs = __builtin_amdgcn_readfirstlane(hs); he = __builtin_amdgcn_readfirstlane(he);
    ws = __builtin_amdgcn_readfirstlane(ws); we = __builtin_amdgcn_readfirstlane(we);
    if constexpr ((ABL & 4) != 0) { hs = (n * 7 + ph) % (H - 3); he = hs + 3; ws = (n * 13 + pw) % (W - 3); we = ws + 3; }
    if constexpr ((ABL & 1) != 0) { he = hs; }
    const int ch = cq * 256 + lane * 4;
    if (he > hs && we > ws && ch < C) {
      m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int y = hs; y < he; ++y) {
        const float *row = pm + ((size_t)y * W + ws) * C + ch;
        for (int x = ws; x < we; ++x, row += C) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(row);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
        }
      }
    }
  }
  stage[wave][lane] = m;
  __syncthreads();
  // 32 channel blocks x (4 rois x 8 floats = 128 contiguous bytes): thread = (channel block, 16-byte piece of the line)
  const int t = threadIdx.x, cbl = t >> 3, j4 = t & 7, roi = j4 >> 1, hf = j4 & 1;
  const int cb = cq * 32 + cbl;
  if (cb < Cb && n0 + roi < N && !((ABL & 2) != 0 && Mp != 7)) {
    const f32x4 v = stage[roi][cbl * 2 + hf];
    *reinterpret_cast<f32x4 *>(xc8 + (((size_t)cb * PH * PW + bin) * Mp + n0) * 8 + j4 * 4) = v;
  }
}

size_t pixel_major_elems(Act feat) { return (size_t)feat.H * feat.W * feat.Cb() * 8; }

int c8p_to_pixel_major(Act feat, float *d_pm, hipStream_t s) {
  MPN_CHECK_ARG(feat.p && d_pm);
  const size_t total = (size_t)feat.H * feat.W * feat.Cb() * 2;
  hipLaunchKernelGGL(c8p_to_pixel_major_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, feat.p, feat.Cb(), feat.H, feat.W, feat.Hp,
                     feat.Wp, d_pm);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

int roi_pool_pm(Act feat, const float *d_pm, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr,
                float *d_x_c8, hipStream_t s, int roi_stride, int Mp) {
  MPN_CHECK_ARG(d_pm && d_rois && d_x_c8 && N > 0 && PH > 0 && PW > 0);
  auto kern = roi_pool_pm_kernel<0>;
#ifdef MPN_DEBUG_HOOKS
  switch (g_gemm_ablate & 7) {
    case 1: kern = roi_pool_pm_kernel<1>; break;
    case 2: kern = roi_pool_pm_kernel<2>; break;
    case 3: kern = roi_pool_pm_kernel<3>; break;
    case 4: kern = roi_pool_pm_kernel<4>; break;
    case 6: kern = roi_pool_pm_kernel<6>; break;
    default: break;
  }
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(N, 4), (unsigned)(PH * PW), (unsigned)cdiv(feat.Cb(), 32)), dim3(256), 0, s, d_pm,
                     feat.Cb(), feat.H, feat.W, d_rois, roi_stride, N, PH, PW, scale, rr, d_x_c8, Mp > 0 ? Mp : lin_mp(N));
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// ---- the range-max-table pooling on PIXEL-MAJOR tables (MultiPathNet head, round 3) -----------------------------------------------
// roi_pool_c8_rmq_kernel's wave is 32 different ROIs at one (channel block, bin): every load instruction touches 32 cache lines and
// uses a quarter of each (115 us per 100 MB pooled matrix = 0.9 TB/s, a sixth of the write ceiling; 11 such pools per MultiPathNet
// image).  Here the map and its vertical range-max levels are pixel-major ([level][y][x][C], level 0 = the map itself), a wave is ONE
// (roi, bin) and its lanes are 256 consecutive channels — every load is one contiguous 1 KiB — the window is wave-uniform, four
// consecutive ROIs share a block and leave as whole 128-byte lines of the mix GEMM's operand (as roi_pool_pm_kernel).  SS: the
// block also emits this (roi, bin, 256-channel slice)'s sum of squares (a fixed shuffle tree), so nn.Normalize's reduction costs a
// PP * C/256-term finish per ROI instead of another pass over the 100 MB matrix (l2norm_partial_kernel).  max is exact: the pooled
// values are bit-identical to the direct and to the C8P range-max kernels.
__global__ void vmax_level_pm_kernel(const f32x4 *__restrict__ prev, f32x4 *__restrict__ out, int H, int W, int C4, int step) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = (size_t)W * C4, total = (size_t)H * row;
  if (t >= total) return;
  const int y = (int)(t / row);
  f32x4 a = prev[t];
  if (y + step < H) {
    const f32x4 b = prev[t + (size_t)step * row];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = b[e] > a[e] ? b[e] : a[e];
  }
  out[t] = a;
}

int build_vmax_tables_pm(Act feat, float *d_tables, hipStream_t s) {  // levels 0 .. vmax_levels_for(H), pixel_major_elems(feat) floats each
  MPN_CHECK_ARG(feat.p && d_tables);
  int rc = c8p_to_pixel_major(feat, d_tables, s);
  if (rc) return rc;
  const int L = vmax_levels_for(feat.H);
  const size_t lvl = pixel_major_elems(feat), total = lvl / 4;
  for (int k = 1; k <= L; ++k) {
    hipLaunchKernelGGL(vmax_level_pm_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, reinterpret_cast<const f32x4 *>(d_tables + (size_t)(k - 1) * lvl),
                       reinterpret_cast<f32x4 *>(d_tables + (size_t)k * lvl), feat.H, feat.W, feat.Cb() * 2, 1 << (k - 1));
    MPN_CHECK_LAUNCH();
  }
  return MPN_OK;
}

// one wave per ROI: its G partial sums of squares (contiguous) in a fixed lane-strided + tree order -> sqrt(sum + 1e-10)
__global__ __launch_bounds__(256) void l2norm_finish_rows_kernel(const float *__restrict__ part, int G, int N, float *__restrict__ nrm) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float ss = 0.0f;
  for (int g = lane; g < G; g += 64) ss += part[(size_t)n * G + g];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) nrm[n] = sqrtf(ss + 1e-10f);
}

template <bool SS>
__global__ __launch_bounds__(256) void roi_pool_pm_rmq_kernel(const float *__restrict__ tab, size_t level_elems, int Cb, int H, int W,
                                                              const float *__restrict__ rois, int roi_stride, int N, int PH, int PW, float scale,
                                                              RoiRule rr, float *__restrict__ xc8, int Mp,
                                                              float *__restrict__ ss_part) {
  __shared__ f32x4 stage[4][64];   // [roi of the quad][lane] = 4 channels
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 4, n = n0 + wave;
  const int bin = blockIdx.y, cq = blockIdx.z;   // cq: which 256-channel slice
  const int ph = bin / PW, pw = bin - ph * PW;
  const int C = Cb * 8;
  f32x4 m = f32x4{0, 0, 0, 0};
  if (n < N) {
    const float *ro = rois + (size_t)roi_stride * n;
    int hs, he, ws, we;
    roi_bin_bounds(ro, scale, rr, H, W, PH, PW, ph, pw, hs, he, ws, we);
    hs = __builtin_amdgcn_readfirstlane(hs); he = __builtin_amdgcn_readfirstlane(he);
    ws = __builtin_amdgcn_readfirstlane(ws); we = __builtin_amdgcn_readfirstlane(we);
    const int ch = cq * 256 + lane * 4;
    if (he > hs && we > ws && ch < C) {
      const int k = 31 - __clz(he - hs);  // 2^k <= he - hs < 2^(k+1)
      const float *lvl = tab + (size_t)k * level_elems + ch;
      const float *r0 = lvl + (size_t)hs * W * C, *r1 = lvl + (size_t)(he - (1 << k)) * W * C;
      m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int xb = ws; xb < we; xb += 4) {  // 8 independent 1-KiB wave loads in flight; positions past the window re-read its last column
        f32x4 a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const size_t xo = (size_t)min(xb + j, we - 1) * C;
          a[j] = *reinterpret_cast<const f32x4 *>(r0 + xo);
          b[j] = *reinterpret_cast<const f32x4 *>(r1 + xo);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = b[j][e] > a[j][e] ? b[j][e] : a[j][e];
            if (v > m[e]) m[e] = v;
          }
      }
    }
  }
  if constexpr (SS) {  // sum of squares of this (roi, bin, slice): lanes in a fixed tree (channels past C contribute +0)
    float ss = m[0] * m[0];
    ss += m[1] * m[1]; ss += m[2] * m[2]; ss += m[3] * m[3];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0 && n < N) ss_part[(size_t)n * (gridDim.y * gridDim.z) + bin * gridDim.z + cq] = ss;  // [roi][bin][slice]
  }
  stage[wave][lane] = m;
  __syncthreads();
  // 32 channel blocks x (4 rois x 8 floats = 128 contiguous bytes): thread = (channel block, 16-byte piece of the line)
  const int t = threadIdx.x, cbl = t >> 3, j4 = t & 7, roi = j4 >> 1, hf = j4 & 1;
  const int cb = cq * 32 + cbl;
  if (cb < Cb && n0 + roi < N) {
    const f32x4 v = stage[roi][cbl * 2 + hf];
    *reinterpret_cast<f32x4 *>(xc8 + (((size_t)cb * PH * PW + bin) * Mp + n0) * 8 + j4 * 4) = v;
  }
}

// pool (+ optionally nn.Normalize(2) x mul, or nn.MulConstant(mul)) one map into its channel range of the mix GEMM's operand
// per-ROI scale mul / sqrt(sum + 1e-10) for n < N, 0 for the padding rows N .. Mp-1 (their pooled rows are zero)
__global__ __launch_bounds__(256) void l2norm_scale_rows_kernel(const float *__restrict__ part, int G, int N, int Mp, float mul, float *__restrict__ scale) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= Mp) return;
  float ss = 0.0f;
  if (n < N) for (int g = lane; g < G; g += 64) ss += part[(size_t)n * G + g];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
  if (lane == 0) scale[n] = n < N ? mul / sqrtf(ss + 1e-10f) : 0.0f;
}

int roi_pool_pm_rmq(Act feat, const float *d_tables_pm, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr, float *d_x_c8, hipStream_t s, int roi_stride, int Mp, int normalize, float mul, float *d_scale_out) {
  MPN_CHECK_ARG(feat.p && d_tables_pm && d_rois && d_x_c8 && N > 0 && PH > 0 && PW > 0);
  const int mp = Mp > 0 ? Mp : lin_mp(N), Cq = cdiv(feat.Cb(), 32), PP = PH * PW;
  const dim3 grid((unsigned)cdiv(N, 4), (unsigned)PP, (unsigned)Cq);
  if (!normalize) {
    hipLaunchKernelGGL(roi_pool_pm_rmq_kernel<false>, grid, dim3(256), 0, s, d_tables_pm, pixel_major_elems(feat), feat.Cb(), feat.H, feat.W, d_rois,
                       roi_stride, N, PH, PW, scale, rr, d_x_c8, mp, (float *)nullptr);
    MPN_CHECK_LAUNCH();
    return mul_const_c8(d_x_c8, feat.Cb() * PP, mp, N, mul, s);
  }
  const int G = PP * Cq;
  void *ws = nullptr;
  { int rc_ws = scratch_get(SCR_L2NORM, ((size_t)G + 1) * N * sizeof(float), s, &ws); if (rc_ws) return rc_ws; }
  float *part = static_cast<float *>(ws), *nrm = part + (size_t)G * N;
  hipLaunchKernelGGL(roi_pool_pm_rmq_kernel<true>, grid, dim3(256), 0, s, d_tables_pm, pixel_major_elems(feat), feat.Cb(), feat.H, feat.W, d_rois,
                     roi_stride, N, PH, PW, scale, rr, d_x_c8, mp, part);
  MPN_CHECK_LAUNCH();
  if (d_scale_out) {  // the consumer GEMM applies the scale (linear_c8_rowscaled): the pooled matrix is left as pooled
    hipLaunchKernelGGL(l2norm_scale_rows_kernel, dim3(cdiv(mp, 4)), dim3(256), 0, s, part, G, N, mp, mul, d_scale_out);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  }
  hipLaunchKernelGGL(l2norm_finish_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, part, G, N, nrm);
  MPN_CHECK_LAUNCH();
  const size_t total = (size_t)feat.Cb() * PP * N * 2;
  hipLaunchKernelGGL(l2norm_apply_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, d_x_c8, (size_t)feat.Cb() * PP, mp, N, nrm, mul);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}


int roi_pool_c8(Act feat, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr,
                float *d_x_c8, int32_t *d_argmax, hipStream_t s, int roi_stride, int Mp) {
  MPN_CHECK_ARG(feat.p && d_rois && d_x_c8 && N > 0 && PH > 0 && PW > 0);
  size_t total = (size_t)feat.Cb() * PH * PW * N * 2;
  hipLaunchKernelGGL(roi_pool_c8_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, feat.p, feat.C, feat.H, feat.W, feat.Hp,
                     feat.Wp, d_rois, roi_stride, N, PH, PW, scale, rr, d_x_c8, Mp > 0 ? Mp : lin_mp(N), d_argmax);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

}  // namespace mpn

using namespace mpn;

// ---- test / bench hooks (not part of the reference surface; DEBUG flavour of the library only) ------
#ifdef MPN_DEBUG_HOOKS
extern "C" void mpn_debug_set_wino_trace(void *p) { g_wino_trace = static_cast<unsigned long long *>(p); }
extern "C" void mpn_debug_set_conv_variant(int v) { g_conv_variant = v; }
extern "C" void mpn_debug_set_wino_tc(int v) { g_wino_tc = v; }
extern "C" void mpn_debug_set_conv_split(int v) { g_conv_split = v; }
extern "C" void mpn_debug_set_gemm_split(int v) { g_gemm_split = v; }
extern "C" void mpn_debug_set_gemm_rsi(int v) { g_gemm_rsi = v; }
extern "C" void mpn_debug_set_split3_ranges(int v) { g_split3_ranges = v; }
extern "C" void mpn_debug_set_gemm_kch(int v) { g_gemm_kch = (v == 4 || v == 8) ? v : 0; }
extern "C" void mpn_debug_set_gemm_ablate(int v) { g_gemm_ablate = v; }

// Kernel-only timing of one conv layer / one linear layer in the pipeline's own layouts (tools/bench_layers.py).
extern "C" int mpn_debug_bench_conv(int Cin, int Cout, int H, int W, int pool, int iters, float *ms_out) {
  MPN_CHECK_ARG(Cin > 0 && Cout > 0 && H > 0 && W > 0 && iters > 0 && ms_out);
  float *in = nullptr, *out = nullptr, *pl = nullptr, *wpk = nullptr, *bpk = nullptr;
  MPN_CHECK_HIP(hipMalloc(&in, act_bytes(Cin, H, W)));
  MPN_CHECK_HIP(hipMalloc(&out, act_bytes(Cout, H, W)));
  MPN_CHECK_HIP(hipMalloc(&pl, act_bytes(Cout, (H + 1) / 2, (W + 1) / 2)));
  MPN_CHECK_HIP(hipMalloc(&wpk, conv_wpk_elems(Cin, Cout) * sizeof(float)));
  MPN_CHECK_HIP(hipMalloc(&bpk, conv_coutp(Cout) * sizeof(float)));
  float *wino = nullptr;
  MPN_CHECK_HIP(hipMalloc(&wino, conv_wino_elems(Cin, Cout) * sizeof(float)));
  MPN_CHECK_HIP(hipMemset(in, 0, act_bytes(Cin, H, W)));
  MPN_CHECK_HIP(hipMemset(out, 0, act_bytes(Cout, H, W)));
  MPN_CHECK_HIP(hipMemset(pl, 0, act_bytes(Cout, (H + 1) / 2, (W + 1) / 2)));
  // non-trivial data (DVFS: zero operands clock higher, guide §5.4 rule 25)
  {
    size_t n = act_bytes(Cin, H, W) / 4, nw = conv_wpk_elems(Cin, Cout);
    std::vector<float> h(n > nw ? n : nw);
    unsigned x = 12345u;
    const char *fill = getenv("MPN_BENCH_FILL");  // "zero" / "const": DVFS / power experiments
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    if (fill && fill[0] == 'z') for (auto &v : h) v = 0.0f;
    if (fill && fill[0] == 'c') for (auto &v : h) v = 0.5f;
    Act ai = make_act(in, Cin, H, W);
    std::vector<float> hi(n, 0.0f);
    for (int cb = 0; cb < ai.Cb(); ++cb)
      for (int y = 0; y < H; ++y)
        for (int xx = 0; xx < W; ++xx)
          for (int j = 0; j < 8; ++j)
            if (cb * 8 + j < Cin) hi[(((size_t)cb * ai.Hp + y + 1) * ai.Wp + xx + 1) * 8 + j] = h[((size_t)(cb * 8 + j) * H + y) % n];
    MPN_CHECK_HIP(hipMemcpy(in, hi.data(), n * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw; ++i) h[i] *= 0.05f;
    MPN_CHECK_HIP(hipMemcpy(wpk, h.data(), nw * 4, hipMemcpyHostToDevice));
    MPN_CHECK_HIP(hipMemset(bpk, 0, conv_coutp(Cout) * sizeof(float)));
    std::vector<float> hw(conv_wino_elems(Cin, Cout));
    for (auto &v : hw) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * 0.05f; }
    if (fill && fill[0] == 'z') for (auto &v : hw) v = 0.0f;
    MPN_CHECK_HIP(hipMemcpy(wino, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  }
  Act ai = make_act(in, Cin, H, W), ao = make_act(out, Cout, H, W), ap = make_act(pl, Cout, (H + 1) / 2, (W + 1) / 2);
  hipEvent_t e0, e1;
  MPN_CHECK_HIP(hipEventCreate(&e0));
  MPN_CHECK_HIP(hipEventCreate(&e1));
  int rc = MPN_OK;
  const bool first = Cin <= 4 && !pool;  // the K = 36 first-layer kernel (its w36 block fits inside the wino allocation)
  auto run = [&]() {
    if (first) return conv3x3_first_c8p(ai, wino, bpk, Cout, 1, ao, nullptr);
    return pool ? conv3x3_c8p(ai, wpk, bpk, Cout, 1, Act{}, ap, nullptr, wino) : conv3x3_c8p(ai, wpk, bpk, Cout, 1, ao, Act{}, nullptr, wino);
  };
  for (int i = 0; i < 2 && rc == MPN_OK; ++i) rc = run();
  MPN_CHECK_HIP(hipDeviceSynchronize());
  MPN_CHECK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == MPN_OK; ++i) rc = run();
  MPN_CHECK_HIP(hipEventRecord(e1, nullptr));
  MPN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MPN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(in); (void)hipFree(out); (void)hipFree(pl); (void)hipFree(wpk); (void)hipFree(bpk); (void)hipFree(wino);
  return rc;
}

// Kernel-only timing of the pixel-major ROI pooling on a 512 x 38 x 63 map (tools/ablate_roipool.py); rois: N x 5 host floats.
extern "C" int mpn_debug_bench_roipool(const float *h_rois, int N, int C, int H, int W, int iters, float *ms_out) {
  MPN_CHECK_ARG(h_rois && N > 0 && C % 8 == 0 && iters > 0 && ms_out);
  Act feat = make_act(nullptr, C, H, W);
  float *pm = nullptr, *rois = nullptr, *x = nullptr;
  const size_t pe = pixel_major_elems(feat), xe = (size_t)(C / 8) * 49 * lin_mp(N) * 8;
  MPN_CHECK_HIP(hipMalloc(&pm, pe * 4)); MPN_CHECK_HIP(hipMalloc(&rois, (size_t)N * 5 * 4)); MPN_CHECK_HIP(hipMalloc(&x, xe * 4));
  {
    std::vector<float> h(pe);
    unsigned s = 99u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) * (1.0f / 16777216.0f); }
    MPN_CHECK_HIP(hipMemcpy(pm, h.data(), pe * 4, hipMemcpyHostToDevice));
    MPN_CHECK_HIP(hipMemcpy(rois, h_rois, (size_t)N * 5 * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  MPN_CHECK_HIP(hipEventCreate(&e0)); MPN_CHECK_HIP(hipEventCreate(&e1));
  int rc = MPN_OK;
  for (int i = 0; i < 2 && rc == MPN_OK; ++i) rc = roi_pool_pm(feat, pm, rois, N, 7, 7, 1.0f / 16, RoiRule{0.0f, 0, 0}, x, nullptr, 5, 0);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  MPN_CHECK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == MPN_OK; ++i) rc = roi_pool_pm(feat, pm, rois, N, 7, 7, 1.0f / 16, RoiRule{0.0f, 0, 0}, x, nullptr, 5, 0);
  MPN_CHECK_HIP(hipEventRecord(e1, nullptr));
  MPN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MPN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(pm); (void)hipFree(rois); (void)hipFree(x);
  return rc;
}

extern "C" int mpn_debug_bench_linear(int M, int K, int N, int iters, float *ms_out) {
  MPN_CHECK_ARG(M > 0 && K > 0 && N > 0 && iters > 0 && ms_out);
  const int K32 = round_up(K, 64);
  size_t xe = mat_c8_elems(M, K32), we = lin_wpk_elems(K32, N), ye = (size_t)(lin_np(N) / 8) * lin_mp(M) * 8;
  float *x = nullptr, *w = nullptr, *b = nullptr, *y = nullptr;
  MPN_CHECK_HIP(hipMalloc(&x, xe * 4)); MPN_CHECK_HIP(hipMalloc(&w, we * 4));
  MPN_CHECK_HIP(hipMalloc(&b, lin_np(N) * 4)); MPN_CHECK_HIP(hipMalloc(&y, ye * 4));
  {
    std::vector<float> h(xe > we ? xe : we);
    unsigned s = 777u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    MPN_CHECK_HIP(hipMemcpy(x, h.data(), xe * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < we; ++i) h[i] *= 0.02f;
    MPN_CHECK_HIP(hipMemcpy(w, h.data(), we * 4, hipMemcpyHostToDevice));
    MPN_CHECK_HIP(hipMemset(b, 0, lin_np(N) * 4));
  }
  hipEvent_t e0, e1;
  MPN_CHECK_HIP(hipEventCreate(&e0)); MPN_CHECK_HIP(hipEventCreate(&e1));
  int rc = MPN_OK;
  for (int i = 0; i < 2 && rc == MPN_OK; ++i) rc = linear_c8(x, M, K, w, b, N, 1, y, nullptr, nullptr);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  MPN_CHECK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == MPN_OK; ++i) rc = linear_c8(x, M, K, w, b, N, 1, y, nullptr, nullptr);
  MPN_CHECK_HIP(hipEventRecord(e1, nullptr));
  MPN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MPN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(y);
  return rc;
}

// test hook: direct vs range-max-table ROI pooling of the same NCHW map; returns the number of output words that differ
__global__ void count_diff_kernel(const float *a, const float *b, size_t n, int *cnt) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n && !(a[t] == b[t])) atomicAdd(cnt, 1);
}
static int g_dbg_roi_bins = 0;  // mpn_debug_set_roi_bins: the bin rule (MPN_ROI_BINS_*) of mpn_debug_roi_pool_rmq_mismatches
extern "C" void mpn_debug_set_roi_bins(int v) { g_dbg_roi_bins = v; }
extern "C" int mpn_debug_roi_pool_rmq_mismatches(const float *d_feat_nchw, int C, int H, int W, const float *d_rois, int roi_stride, int N,
                                                 int PH, int PW, float scale, int *n_mismatch) {
  MPN_CHECK_ARG(d_feat_nchw && d_rois && n_mismatch && C > 0 && H > 0 && W > 0 && N > 0);
  const RoiRule rr{1.0f, 0, g_dbg_roi_bins};
  float *act = nullptr, *tab = nullptr, *o1 = nullptr, *o2 = nullptr;
  int *cnt = nullptr;
  const size_t ab = act_bytes(C, H, W);
  const int L = vmax_levels_for(H);
  const size_t oe = (size_t)((C + 7) / 8) * PH * PW * lin_mp(N) * 8;
  MPN_CHECK_HIP(hipMalloc(&act, ab));
  MPN_CHECK_HIP(hipMalloc(&tab, ab * (L > 0 ? L : 1)));
  MPN_CHECK_HIP(hipMalloc(&o1, oe * 4)); MPN_CHECK_HIP(hipMalloc(&o2, oe * 4));
  MPN_CHECK_HIP(hipMalloc(&cnt, 4));
  MPN_CHECK_HIP(hipMemset(act, 0, ab)); MPN_CHECK_HIP(hipMemset(o1, 0, oe * 4)); MPN_CHECK_HIP(hipMemset(o2, 0, oe * 4));
  MPN_CHECK_HIP(hipMemset(cnt, 0, 4));
  Act a = make_act(act, C, H, W);
  int rc = nchw_to_c8p(d_feat_nchw, C, H, W, a, nullptr);
  if (rc == MPN_OK) rc = build_vmax_tables(a, tab, nullptr);
  if (rc == MPN_OK) rc = roi_pool_c8(a, d_rois, N, PH, PW, scale, rr, o1, nullptr, nullptr, roi_stride, 0);
  if (rc == MPN_OK) rc = roi_pool_c8_rmq(a, tab, d_rois, N, PH, PW, scale, rr, o2, nullptr, roi_stride, 0);
  if (rc == MPN_OK) {
    hipLaunchKernelGGL(count_diff_kernel, dim3((unsigned)cdiv_sz(oe, 256)), dim3(256), 0, nullptr, o1, o2, oe, cnt);
    MPN_CHECK_LAUNCH();
  }
  // the pixel-major range-max path (round 3): also bit-identical to the direct kernel
  float *tabpm = nullptr;
  if (rc == MPN_OK) {
    MPN_CHECK_HIP(hipMalloc(&tabpm, pixel_major_elems(a) * sizeof(float) * (L + 1)));
    MPN_CHECK_HIP(hipMemset(o2, 0, oe * 4));
    rc = build_vmax_tables_pm(a, tabpm, nullptr);
    if (rc == MPN_OK) rc = roi_pool_pm_rmq(a, tabpm, d_rois, N, PH, PW, scale, rr, o2, nullptr, roi_stride, 0, 0, 1.0f, nullptr);
    if (rc == MPN_OK) {
      hipLaunchKernelGGL(count_diff_kernel, dim3((unsigned)cdiv_sz(oe, 256)), dim3(256), 0, nullptr, o1, o2, oe, cnt);
      MPN_CHECK_LAUNCH();
    }
  }
  if (rc == MPN_OK) MPN_CHECK_HIP(hipMemcpy(n_mismatch, cnt, 4, hipMemcpyDeviceToHost));
  (void)hipFree(act); (void)hipFree(tab); (void)hipFree(o1); (void)hipFree(o2); (void)hipFree(cnt);
  if (tabpm) (void)hipFree(tabpm);
  return rc;
}

#endif  // MPN_DEBUG_HOOKS

// ---- module-level C entry points (NCHW / row-major Torch layouts) -------------------------------
extern "C" size_t mpn_conv3x3_workspace_bytes(int B, int Cin, int H, int W, int Cout) {
  (void)B;
  size_t a = act_bytes(Cin, H, W), o = act_bytes(Cout, H, W);
  size_t w = (conv_wpk_elems(Cin, Cout) + conv_wino_elems(Cin, Cout) + (size_t)conv_coutp(Cout)) * sizeof(float);
  return a + o + w + 1024;
}

extern "C" int mpn_conv3x3_forward(const float *d_in, int B, int Cin, int H, int W, const float *d_w, const float *d_b,
                                   int Cout, int relu, float *d_out, void *d_ws, size_t ws_bytes, void *stream) {
  MPN_CHECK_ARG(d_in && d_w && d_out && d_ws && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0);
  if (ws_bytes < mpn_conv3x3_workspace_bytes(B, Cin, H, W, Cout)) {
    set_error("mpn_conv3x3_forward: workspace too small (%zu < %zu)", ws_bytes, mpn_conv3x3_workspace_bytes(B, Cin, H, W, Cout));
    return MPN_ENOMEM;
  }
  hipStream_t s = as_stream(stream);
  char *ws = static_cast<char *>(d_ws);
  size_t ab = act_bytes(Cin, H, W), ob = act_bytes(Cout, H, W);
  Act ain = make_act(reinterpret_cast<float *>(ws), Cin, H, W);
  Act aout = make_act(reinterpret_cast<float *>(ws + ab), Cout, H, W);
  float *wpk = reinterpret_cast<float *>(ws + ab + ob);
  float *bpk = wpk + conv_wpk_elems(Cin, Cout);
  float *wino = bpk + conv_coutp(Cout);
  int rc = pack_conv_weights(d_w, d_b, Cin, Cout, wpk, bpk, s);
  if (rc) return rc;
  rc = pack_conv_weights_wino(d_w, Cin, Cout, wino, s);
  if (rc) return rc;
  for (int b = 0; b < B; ++b) {
    MPN_CHECK_HIP(hipMemsetAsync(ain.p, 0, ab, s));  // zero halo (+ pad channels)
    rc = nchw_to_c8p(d_in + (size_t)b * Cin * H * W, Cin, H, W, ain, s);
    if (rc) return rc;
    rc = conv3x3_c8p(ain, wpk, bpk, Cout, relu, aout, Act{}, s, wino);
    if (rc) return rc;
    rc = c8p_to_nchw(aout, d_out + (size_t)b * Cout * H * W, s);
    if (rc) return rc;
  }
  return MPN_OK;
}

extern "C" int mpn_maxpool2x2_ceil_forward(const float *d_in, int BC, int H, int W, float *d_out, void *stream) {
  MPN_CHECK_ARG(d_in && d_out && BC > 0 && H > 0 && W > 0);
  size_t total = (size_t)BC * ((H + 1) / 2) * ((W + 1) / 2);
  hipLaunchKernelGGL(maxpool2x2_nchw_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, as_stream(stream), d_in, (size_t)BC, H, W, d_out);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// nn.Linear on Torch row-major tensors: packs W and x into the C8 layouts in a cached scratch
// allocation (grown on demand), runs the MFMA GEMM, writes row-major y.
extern "C" int mpn_linear_forward(const float *d_x, int M, int K, const float *d_w, const float *d_b, int N, int relu,
                                  float *d_y, void *stream) {
  MPN_CHECK_ARG(d_x && d_w && d_y && M > 0 && K > 0 && N > 0);
  hipStream_t s = as_stream(stream);
  size_t xe = mat_c8_elems(M, round_up(K, 64)), we = lin_wpk_elems(round_up(K, 64), N), be = lin_np(N);
  size_t need = (xe + we + be) * sizeof(float);
  void *ws = nullptr;
  { int rc_ws = scratch_get(SCR_LINEAR_PACK, need, s, &ws); if (rc_ws) return rc_ws; }
  float *xc8 = static_cast<float *>(ws), *wpk = xc8 + xe, *bpk = wpk + we;
  MPN_CHECK_HIP(hipMemsetAsync(xc8, 0, xe * sizeof(float), s));
  int rc = rowmajor_to_c8(d_x, M, K, xc8, s);
  if (rc) return rc;
  rc = pack_linear_weights(d_w, d_b, K, N, 1, wpk, bpk, s);
  if (rc) return rc;
  return linear_c8(xc8, M, K, wpk, bpk, N, relu, nullptr, d_y, s);
}
