// probe: how much side work fits in the 64-cycle shadow of one v_mfma_f32_32x32x2_f32 when the SAME wave issues it
// (one wave per SIMD, as in the Winograd kernel)?  Loop of 16 MFMAs (4 accumulators); N copies of instruction X after each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int N, int NT> __global__ __launch_bounds__(NT) void k(const float* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x * 8 + i) * 2]; b[i] = src[(threadIdx.x * 8 + i) * 2 + 1]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  f32x2 v0 = {a[0], a[1]}, v1 = {b[0], b[1]}, v2 = {a[2], b[2]};
  f32x4 r4 = {0, 0, 0, 0};
  float s0 = a[3], s1 = b[3];
  float* lw = lds + threadIdx.x * 2;            // conflict-free 8-byte slots
  const float* lr = lds + threadIdx.x * 4;      // conflict-free 16-byte slots
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      __builtin_amdgcn_sched_barrier(0);
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m & 7], b[(m + 1) & 7], acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        if (KIND == 0) { s0 = s0 + s1; s1 = s1 - s0; }                 // 2 dependent v_add_f32
        if (KIND == 1) { v0 = v0 + v1; v1 = v1 - v2; }                 // 2 v_pk_add_f32
        if (KIND == 2) *reinterpret_cast<f32x2*>(lw + ((m * 4 + n) & 15) * 512) = v0;           // ds_write_b64
        if (KIND == 3) r4 += *reinterpret_cast<const f32x4*>(lr + ((m * 4 + n) & 7) * 1024);   // ds_read_b128 (+ use)
        if (KIND == 4) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (threadIdx.x & 63) * 4),
                                                        (__attribute__((address_space(3))) void*)(lds + 8192 + wave * 256), 16, 0, 0);
      }
    }
    if (KIND == 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  float s = s0 + s1 + v0[0] + v0[1] + v1[0] + v1[1] + r4[0] + r4[1] + r4[2] + r4[3];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
}
template <int KIND, int N, int NT> void run(const float* d, float* o, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = k<KIND, N, NT>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    int iters = 4000;
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(256), dim3(NT), 65536, 0, d, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-22s x%d per MFMA, %d waves/SIMD: %.1f TF/s  (%.1f ns per MFMA per SIMD)\n", name, N, NT / 256, 256.0 * (NT / 64) * iters * 16 * 4096.0 / ms / 1e9, ms * 1e6 / (iters * 16.0 * (NT / 256)));
  }
}
#define RUNS(K, NAME) run<K, 0, 256>(d, o, NAME); run<K, 1, 256>(d, o, NAME); run<K, 2, 256>(d, o, NAME); run<K, 4, 256>(d, o, NAME); run<K, 8, 256>(d, o, NAME); \
  run<K, 0, 512>(d, o, NAME); run<K, 1, 512>(d, o, NAME); run<K, 2, 512>(d, o, NAME); run<K, 4, 512>(d, o, NAME); run<K, 8, 512>(d, o, NAME);
int main() {
  std::vector<float> h(256 * 16 + 1024); for (auto& v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
  float *d, *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 256 * 512 * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  RUNS(0, "2x v_add_f32") RUNS(1, "2x v_pk_add_f32") RUNS(2, "ds_write_b64")
  return 0;
}
