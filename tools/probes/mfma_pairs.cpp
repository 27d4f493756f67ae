// probe: the Winograd kernel's MFMA issue pattern in isolation — NACC accumulators, 8 MFMAs per accumulator pair
// alternating between its two accumulators (dependency distance 2), vs the same count with distance-4 interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int GROUP>
__global__ __launch_bounds__(256) void k(const float* src, float* out, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) * 2]; b[i] = src[(threadIdx.x * 8 + i) * 2 + 1]; }
  f32x16 acc[NACC];
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep)
#pragma unroll
      for (int p = 0; p < NACC / GROUP; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < GROUP; ++g)
            acc[GROUP * p + g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + g) & 7], b[(j + 4 + g) & 7], acc[GROUP * p + g], 0, 0, 0);
  }
  float s = 0; for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int GROUP> void run(const float* d, float* o) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    int iters = 5000;
    hipEventRecord(e0); hipLaunchKernelGGL((k<NACC, GROUP>), dim3(256), dim3(256), 0, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("nacc=%2d group=%d: %.1f TF/s\n", NACC, GROUP, 256.0 * 4 * iters * 64 * (2.0 * 32 * 32 * 2) / ms / 1e9);
  }
}
int main() {
  std::vector<float> h(256 * 16); for (auto& v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
  float *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 256 * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<2, 2>(d, o); run<4, 2>(d, o); run<4, 4>(d, o); run<8, 2>(d, o); run<8, 4>(d, o); run<16, 2>(d, o); run<16, 4>(d, o); run<16, 8>(d, o);
  return 0;
}
