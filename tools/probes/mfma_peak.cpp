// probe: sustained v_mfma_f32_32x32x2_f32 rate with (a) constant and (b) random per-lane operands (DVFS / power)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int BLOCKS_PER_CU>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) * 2]; b[i] = src[(threadIdx.x * 8 + i) * 2 + 1]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[(i + 1) & 7], acc[3], 0, 0, 0);
    }
  }
  float s = 0; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  const int iters = 20000;
  float *src, *out; hipMalloc(&src, 256 * 16 * 4); hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> h(256 * 16);
    for (auto& v : h) v = mode == 0 ? 0.f : mode == 1 ? 0.5f : (rand() / (float)RAND_MAX * 2.f - 1.f);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, src, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 256.0 * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
      if (rep) printf("mode %s: %.2f ms  %.1f TFLOP/s\n", mode == 0 ? "zero" : mode == 1 ? "const" : "random", ms, fl / ms / 1e9);
    }
  }
}
