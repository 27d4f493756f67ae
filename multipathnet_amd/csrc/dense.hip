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

// =================================================================================================
// conv3x3, stride 1, pad 1, C8P in/out
// =================================================================================================
struct ConvArgs {
  const float *in; size_t in_plane; int in_Wp;
  const float *wpk; int CoutP; const float *bpk;
  float *out; size_t out_plane; int out_Wp;
  float *pool; size_t pool_plane; int pool_Wp; int pool_H, pool_W;
  int H, W, nchunks, out_cb, relu, n_ct, tiles_x;
};

template <int BM, int TH, int WM, int WN, bool GLDS>
__global__ __launch_bounds__(256) void conv3x3_c8p_kernel(ConvArgs a) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int MI = BM / WM / 32, NI = TH / WN;
  static_assert(MI >= 1 && NI >= 1, "tile");
  constexpr int IN_PIECES = (TH + 2) * 68;         // 16-byte pieces of the (TH+2) x 34 px halo tile
  constexpr int IN_LOADS = (IN_PIECES + 63) / 64;  // 1 KiB wave-loads
  constexpr int IN_FLOATS = IN_LOADS * 256;
  constexpr int W_LOADS = 9 * BM / 32;
  constexpr int W_FLOATS = 9 * BM * 8;
  constexpr int STAGE = IN_FLOATS + W_FLOATS;
  constexpr int IN_IT = (IN_LOADS + 3) / 4, W_IT = (W_LOADS + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];

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
    int tap = p / (BM * 2), rem = p - tap * (BM * 2);
    w_off[i] = (tap * a.CoutP + cout0) * 8 + rem * 4;
  }
  const size_t w_chunk = (size_t)9 * a.CoutP * 8;

  f32x4 rin[GLDS ? 1 : IN_IT], rw[GLDS ? 1 : W_IT];
  auto issue = [&](int c, int s) {
    const float *ib = a.in + (size_t)c * a.in_plane;
    const float *wb = a.wpk + (size_t)c * w_chunk;
    float *st = lds + s * STAGE;
#pragma unroll
    for (int i = 0; i < IN_IT; ++i) {
      if constexpr (GLDS) {
        if (in_ok[i]) glds16(ib + in_off[i], st + (i * 4 + wave) * 256);
      } else {
        if (in_ok[i]) rin[i] = *reinterpret_cast<const f32x4 *>(ib + in_off[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      if constexpr (GLDS) {
        if (w_ok[i]) glds16(wb + w_off[i], st + IN_FLOATS + (i * 4 + wave) * 256);
      } else {
        if (w_ok[i]) rw[i] = *reinterpret_cast<const f32x4 *>(wb + w_off[i]);
      }
    }
  };
  auto commit = [&](int s) {  // register-staged path only
    if constexpr (!GLDS) {
      float *st = lds + s * STAGE;
#pragma unroll
      for (int i = 0; i < IN_IT; ++i)
        if (in_ok[i]) *reinterpret_cast<f32x4 *>(st + (i * 4 + wave) * 256 + lane * 4) = rin[i];
#pragma unroll
      for (int i = 0; i < W_IT; ++i)
        if (w_ok[i]) *reinterpret_cast<f32x4 *>(st + IN_FLOATS + (i * 4 + wave) * 256 + lane * 4) = rw[i];
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
  issue(0, 0);
  commit(0);
  __syncthreads();

  for (int c = 0; c < a.nchunks; ++c) {
    const int s = c & 1;
    if (c + 1 < a.nchunks) issue(c + 1, s ^ 1);
    const float *Il = lds + s * STAGE + lane_off;
    const float *Wl = lds + s * STAGE + IN_FLOATS + lane_off;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4 *>(Wl + (tap * BM + mbase + mi * 32) * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = *reinterpret_cast<const f32x4 *>(Il + ((rbase + ni + dy) * 34 + dx) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
    if (c + 1 < a.nchunks) commit(s ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias + ReLU, C8P float4 stores, optional fused ceil-mode 2x2 max-pool
  const int x = x0 + l31;
  const bool xok = x < a.W;
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
      if constexpr (NI == 2) {
        if (a.pool) {  // rows (y0+rbase, y0+rbase+1) are a vertical pooling pair; lanes (2j,2j+1) a horizontal one
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = fmaxf(v[0][e], v[1][e]);
            m[e] = fmaxf(t, __shfl_xor(t, 1));
          }
          const int py = (y0 + rbase) >> 1, px = x >> 1;
          if (!(l31 & 1) && py < a.pool_H && px < a.pool_W)
            *reinterpret_cast<f32x4 *>(a.pool + (size_t)cb * a.pool_plane + ((size_t)(py + 1) * a.pool_Wp + px + 1) * 8 + half * 4) = m;
        }
      }
    }
  }
}

template <int BM, int TH, int WM, int WN, bool GLDS>
static int launch_conv(const ConvArgs &a0, int tiles_y, hipStream_t s) {
  ConvArgs a = a0;
  constexpr int IN_LOADS = ((TH + 2) * 68 + 63) / 64;
  constexpr size_t LDS = (size_t)2 * (IN_LOADS * 256 + 9 * BM * 8) * sizeof(float);
  auto kern = conv3x3_c8p_kernel<BM, TH, WM, WN, GLDS>;
  static bool attr = false;
  if (!attr) {
    MPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr = true;
  }
  dim3 grid((unsigned)(a.n_ct * tiles_y * a.tiles_x));
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, a);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

static int g_conv_variant = 0;  // 0 = auto; test/bench hook: 1 = force 128x4 tile, 2 = force 64x8, +16 = register staging

int conv3x3_variant_for(int Cout) {
  int variant = g_conv_variant & 15;
  if (variant == 0) variant = (Cout <= 64) ? 2 : 1;
  return variant;
}

int conv3x3_c8p(Act in, const float *d_wpk, const float *d_bpk, int Cout, int relu, Act out, Act pooled, hipStream_t s) {
  MPN_CHECK_ARG(in.p && d_wpk && d_bpk && (out.p || pooled.p));
  ConvArgs a{};
  a.in = in.p; a.in_plane = in.plane(); a.in_Wp = in.Wp;
  a.wpk = d_wpk; a.CoutP = conv_coutp(Cout); a.bpk = d_bpk;
  a.out = out.p; a.out_plane = out.p ? out.plane() : 0; a.out_Wp = out.p ? out.Wp : 0;
  a.pool = pooled.p; a.pool_plane = pooled.p ? pooled.plane() : 0; a.pool_Wp = pooled.p ? pooled.Wp : 0;
  a.pool_H = pooled.p ? pooled.H : 0; a.pool_W = pooled.p ? pooled.W : 0;
  a.H = in.H; a.W = in.W; a.nchunks = in.Cb(); a.out_cb = (Cout + 7) / 8; a.relu = relu;
  a.tiles_x = cdiv(in.W, 32);
  if (out.p) MPN_CHECK_ARG(out.H == in.H && out.W == in.W && out.C == Cout);
  if (pooled.p) MPN_CHECK_ARG(pooled.H == (in.H + 1) / 2 && pooled.W == (in.W + 1) / 2 && pooled.C == Cout);
  const int variant = conv3x3_variant_for(Cout);
  const bool regstage = (g_conv_variant & 16) != 0;
  if (variant == 1) {
    a.n_ct = cdiv(Cout, 128);
    int tiles_y = cdiv(in.H, 4);
    return regstage ? launch_conv<128, 4, 2, 2, false>(a, tiles_y, s) : launch_conv<128, 4, 2, 2, true>(a, tiles_y, s);
  } else {
    a.n_ct = cdiv(Cout, 64);
    int tiles_y = cdiv(in.H, 8);
    return regstage ? launch_conv<64, 8, 1, 4, false>(a, tiles_y, s) : launch_conv<64, 8, 1, 4, true>(a, tiles_y, s);
  }
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
};

constexpr int KCH = 4;  // 8-wide K chunks per LDS stage (32 k)

template <bool GLDS>
__global__ __launch_bounds__(256) void gemm_c8_kernel(GemmArgs a) {
  constexpr int OP_FLOATS = KCH * 128 * 8;  // 16 KiB per operand per stage
  constexpr int STAGE = 2 * OP_FLOATS;
  constexpr int LOADS = OP_FLOATS / 256;    // 16 wave-loads per operand
  constexpr int IT = LOADS / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // XCD-aware tile order: blocks b, b+8, ... share an XCD (L2); give each XCD a contiguous run of
  // logical tiles = the m-tiles of one weight panel, so the panel is fetched from HBM once per XCD.
  int b = blockIdx.x;
  const int nb = gridDim.x;
  if ((nb & 7) == 0) b = (b & 7) * (nb >> 3) + (b >> 3);
  const int nt = b / a.n_mt, mt = b - nt * a.n_mt;
  const int n0 = nt * 128, m0 = mt * 128;
  const int split = blockIdx.y;
  const int st0 = split * a.stages_per_split;
  const int st1 = min(a.nstages, st0 + a.stages_per_split);
  const int wm = wave >> 1, wn = wave & 1;

  int a_off[IT], b_off[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int p = (i * 4 + wave) * 64 + lane;  // 16-byte piece inside the [KCH][128][8] operand tile
    int kk = p >> 8, rem = p & 255;
    a_off[i] = (kk * a.NP + n0) * 8 + rem * 4;
    b_off[i] = (kk * a.Mp + m0) * 8 + rem * 4;
  }
  const size_t a_stage = (size_t)KCH * a.NP * 8, b_stage = (size_t)KCH * a.Mp * 8;

  f32x4 ra[GLDS ? 1 : IT], rb[GLDS ? 1 : IT];
  auto issue = [&](int st, int s) {
    const float *ab = a.wpk + (size_t)st * a_stage;
    const float *bb = a.x + (size_t)st * b_stage;
    float *l = lds + s * STAGE;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      if constexpr (GLDS) {
        glds16(ab + a_off[i], l + (i * 4 + wave) * 256);
        glds16(bb + b_off[i], l + OP_FLOATS + (i * 4 + wave) * 256);
      } else {
        ra[i] = *reinterpret_cast<const f32x4 *>(ab + a_off[i]);
        rb[i] = *reinterpret_cast<const f32x4 *>(bb + b_off[i]);
      }
    }
  };
  auto commit = [&](int s) {
    if constexpr (!GLDS) {
      float *l = lds + s * STAGE;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        *reinterpret_cast<f32x4 *>(l + (i * 4 + wave) * 256 + lane * 4) = ra[i];
        *reinterpret_cast<f32x4 *>(l + OP_FLOATS + (i * 4 + wave) * 256 + lane * 4) = rb[i];
      }
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
  if (st0 < st1) {
    issue(st0, 0);
    commit(0);
  }
  __syncthreads();
  for (int st = st0; st < st1; ++st) {
    const int s = (st - st0) & 1;
    if (st + 1 < st1) issue(st + 1, s ^ 1);
    const float *Al = lds + s * STAGE + lane_off;
    const float *Bl = Al + OP_FLOATS;
#pragma unroll
    for (int kk = 0; kk < KCH; ++kk) {
      f32x4 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[mi] = *reinterpret_cast<const f32x4 *>(Al + (kk * 128 + wm * 64 + mi * 32) * 8);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[ni] = *reinterpret_cast<const f32x4 *>(Bl + (kk * 128 + wn * 64 + ni * 32) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
    if (st + 1 < st1) commit(s ^ 1);
    __syncthreads();
  }

  float *yb = a.y + (a.direct ? (size_t)0 : (size_t)split * (a.NP / 8) * a.Mp * 8);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nb8 = (n0 + wm * 64 + mi * 32) / 8 + g;
      f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.direct) b4 = *reinterpret_cast<const f32x4 *>(a.bpk + nb8 * 8 + half * 4);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int m = m0 + wn * 64 + ni * 32 + l31;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[mi][ni][g * 4 + e] + b4[e];
          if (a.direct && a.relu) t = t < 0.0f ? 0.0f : t;
          v[e] = t;
        }
        if (m < a.M) *reinterpret_cast<f32x4 *>(yb + ((size_t)nb8 * a.Mp + m) * 8 + half * 4) = v;
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

static int g_gemm_regstage = 0;
static float *g_splitk_ws = nullptr;
static size_t g_splitk_ws_bytes = 0;

int linear_c8(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8,
              float *d_y_rm, hipStream_t s) {
  MPN_CHECK_ARG(d_x_c8 && d_wpk && d_bpk && (d_y_c8 || d_y_rm) && M > 0 && K > 0 && N > 0);
  GemmArgs a{};
  a.x = d_x_c8; a.Mp = lin_mp(M); a.wpk = d_wpk; a.NP = lin_np(N); a.bpk = d_bpk;
  a.M = M; a.relu = relu;
  const int K32 = round_up(K, 32);
  a.nstages = K32 / 32;
  a.n_mt = a.Mp / 128; a.n_nt = a.NP / 128;
  const int tiles = a.n_mt * a.n_nt;
  int S = 1;
  if (tiles < 128) {  // too few tiles to fill 256 CUs: split K (deterministic two-pass reduce)
    S = 256 / tiles;
    if (S > a.nstages / 2) S = a.nstages / 2;
    if (S < 1) S = 1;
  }
  a.stages_per_split = cdiv(a.nstages, S);
  S = cdiv(a.nstages, a.stages_per_split);
  const bool direct = (S == 1) && d_y_c8 && !d_y_rm;
  a.direct = direct ? 1 : 0;
  constexpr size_t LDS = (size_t)2 * 2 * KCH * 128 * 8 * sizeof(float);
  static bool attr = false;
  if (!attr) {
    MPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_c8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    MPN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_c8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
    attr = true;
  }
  if (direct) {
    a.y = d_y_c8;
  } else {
    size_t need = (size_t)S * (a.NP / 8) * a.Mp * 8 * sizeof(float);
    if (need > g_splitk_ws_bytes) {  // grows monotonically; steady state allocates nothing
      MPN_CHECK_HIP(hipStreamSynchronize(s));
      if (g_splitk_ws) (void)hipFree(g_splitk_ws);
      g_splitk_ws = nullptr; g_splitk_ws_bytes = 0;
      MPN_CHECK_HIP(hipMalloc(&g_splitk_ws, need));
      g_splitk_ws_bytes = need;
    }
    a.y = g_splitk_ws;
  }
  dim3 grid((unsigned)tiles, (unsigned)S);
  if (g_gemm_regstage) hipLaunchKernelGGL(gemm_c8_kernel<false>, grid, dim3(256), LDS, s, a);
  else hipLaunchKernelGGL(gemm_c8_kernel<true>, grid, dim3(256), LDS, s, a);
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
  int K32 = round_up(K, 32), nq = K32 / 8, NP = lin_np(N);
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
                                                          const float *__restrict__ rois, int N, int PH, int PW, float scale,
                                                          float coord_offset, int end_adjust, float *__restrict__ xc8, int Mp,
                                                          int32_t *__restrict__ argmax) {
  const int Cb = (C + 7) / 8, PP = PH * PW;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)Cb * PP * N * 2;
  if (t >= total) return;
  int h = (int)(t & 1); size_t r = t >> 1;
  int n = (int)(r % N); r /= N;
  int bin = (int)(r % PP); int cb = (int)(r / PP);
  int ph = bin / PW, pw = bin - ph * PW;
  const float *ro = rois + 5 * (size_t)n;
  int sw = (int)roundf((ro[1] - coord_offset) * scale);
  int sh = (int)roundf((ro[2] - coord_offset) * scale);
  int ew = (int)roundf((ro[3] - coord_offset) * scale) + end_adjust;
  int eh = (int)roundf((ro[4] - coord_offset) * scale) + end_adjust;
  int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
  float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
  int hs = (int)floorf((float)ph * bh) + sh, he = (int)ceilf((float)(ph + 1) * bh) + sh;
  int ws = (int)floorf((float)pw * bw) + sw, we = (int)ceilf((float)(pw + 1) * bw) + sw;
  hs = min(max(hs, 0), H); he = min(max(he, 0), H);
  ws = min(max(ws, 0), W); we = min(max(we, 0), W);
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

int roi_pool_c8(Act feat, const float *d_rois, int N, int PH, int PW, float scale, float coord_offset, int end_adjust,
                float *d_x_c8, int32_t *d_argmax, hipStream_t s) {
  MPN_CHECK_ARG(feat.p && d_rois && d_x_c8 && N > 0 && PH > 0 && PW > 0);
  size_t total = (size_t)feat.Cb() * PH * PW * N * 2;
  hipLaunchKernelGGL(roi_pool_c8_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, feat.p, feat.C, feat.H, feat.W, feat.Hp,
                     feat.Wp, d_rois, N, PH, PW, scale, coord_offset, end_adjust, d_x_c8, lin_mp(N), d_argmax);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

}  // namespace mpn

using namespace mpn;

// ---- test / bench hooks (not part of the reference surface) -------------------------------------
extern "C" void mpn_debug_set_conv_variant(int v) { g_conv_variant = v; }
extern "C" void mpn_debug_set_gemm_regstage(int v) { g_gemm_regstage = v; }

// ---- module-level C entry points (NCHW / row-major Torch layouts) -------------------------------
extern "C" size_t mpn_conv3x3_workspace_bytes(int B, int Cin, int H, int W, int Cout) {
  (void)B;
  size_t a = act_bytes(Cin, H, W), o = act_bytes(Cout, H, W);
  size_t w = conv_wpk_elems(Cin, Cout) * sizeof(float) + (size_t)conv_coutp(Cout) * sizeof(float);
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
  int rc = pack_conv_weights(d_w, d_b, Cin, Cout, wpk, bpk, s);
  if (rc) return rc;
  for (int b = 0; b < B; ++b) {
    MPN_CHECK_HIP(hipMemsetAsync(ain.p, 0, ab, s));  // zero halo (+ pad channels)
    rc = nchw_to_c8p(d_in + (size_t)b * Cin * H * W, Cin, H, W, ain, s);
    if (rc) return rc;
    rc = conv3x3_c8p(ain, wpk, bpk, Cout, relu, aout, Act{}, s);
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
  static float *scratch = nullptr;
  static size_t scratch_bytes = 0;
  size_t xe = mat_c8_elems(M, round_up(K, 32)), we = lin_wpk_elems(round_up(K, 32), N), be = lin_np(N);
  size_t need = (xe + we + be) * sizeof(float);
  if (need > scratch_bytes) {
    MPN_CHECK_HIP(hipStreamSynchronize(s));
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr; scratch_bytes = 0;
    MPN_CHECK_HIP(hipMalloc(&scratch, need));
    scratch_bytes = need;
  }
  float *xc8 = scratch, *wpk = scratch + xe, *bpk = wpk + we;
  MPN_CHECK_HIP(hipMemsetAsync(xc8, 0, xe * sizeof(float), s));
  int rc = rowmajor_to_c8(d_x, M, K, xc8, s);
  if (rc) return rc;
  rc = pack_linear_weights(d_w, d_b, K, N, 1, wpk, bpk, s);
  if (rc) return rc;
  return linear_c8(xc8, M, K, wpk, bpk, N, relu, nullptr, d_y, s);
}
