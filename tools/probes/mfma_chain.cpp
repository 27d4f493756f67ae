// probe: v_mfma_f32_32x32x2_f32 rate vs the number of independent accumulator chains (dependency distance)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ __launch_bounds__(256) void k(const float* src, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) * 2]; b[i] = src[(threadIdx.x * 8 + i) * 2 + 1]; }
  f32x16 acc[NCH];
  for (int q = 0; q < NCH; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16 / NCH; ++i)
#pragma unroll
      for (int q = 0; q < NCH; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + q) & 7], b[i & 7], acc[q], 0, 0, 0);
  }
  float s = 0; for (int q = 0; q < NCH; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCH> void run(const float* d, float* o) {
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("chains=%d: %.3f ms  %.1f TF/s\n", NCH, ms, 256.0 * 4 * iters * 16 * (2.0 * 32 * 32 * 2) / ms / 1e9);
  }
}
int main() {
  std::vector<float> h(256 * 16); for (auto& v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
  float *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 256 * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<1>(d, o); run<2>(d, o); run<4>(d, o); run<8>(d, o); run<16>(d, o);
  return 0;
}
