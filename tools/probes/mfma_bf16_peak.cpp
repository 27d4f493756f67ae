// probe: sustained v_mfma_f32_32x32x16_bf16 rate from registers only — 1 or 2 waves per SIMD (grid = 256 or 512 blocks of 4 waves),
// 8 independent accumulators per wave (the <4,2> wave tile of conv2d_c8i_bf16_dma_kernel), zero / random operands, and with
// FILL extra scalar + vector instructions dealt between the MFMAs of every 16-MFMA "stage" (what a stage boundary of the conv loop carries).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int FILL>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ src, float* out, int iters, int dummy) {
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[threadIdx.x * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = src[threadIdx.x * 6 + 4 + i];
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  int sx = dummy, vx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t >> 1], b[t & 1], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (FILL && ks == 0) {  // a "stage boundary": FILL scalar + FILL/2 vector instructions in one lump (not hidden in MFMA shadows)
#pragma unroll
        for (int f = 0; f < FILL; ++f) { sx = sx * 3 + 1; asm volatile("" : "+s"(sx)); if (f & 1) { vx = vx * 5 + sx; asm volatile("" : "+v"(vx)); } }
      }
    }
  }
  float s = (float)(sx + vx); for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int FILL> void run(const bf16x8* src, float* out, int blocks, const char* what) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FILL>, dim3(blocks), dim3(256), 0, 0, src, out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    if (rep) printf("%-8s blocks %4d (%d wave/SIMD) fill %2d: %8.2f ms  %7.1f TFLOP/s  %.3f of 2.5 PF\n", what, blocks, blocks / 256, FILL, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
  }
}
int main() {
  bf16x8* src; float* out; hipMalloc(&src, 256 * 6 * 16); hipMalloc(&out, 512 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    std::vector<unsigned short> h(256 * 6 * 8);
    for (auto& v : h) v = mode == 0 ? 0 : (unsigned short)(0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* w = mode == 0 ? "zero" : "random";
    for (int blocks : {256, 512}) { run<0>(src, out, blocks, w); run<16>(src, out, blocks, w); run<64>(src, out, blocks, w); }
  }
}
