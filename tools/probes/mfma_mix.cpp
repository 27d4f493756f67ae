// probe: which companion traffic costs MFMA throughput?  MODE bits: 1 = ds_read_b128 fragments from LDS per 16 MFMAs,
// 2 = global_load_lds DMA of 32 KiB per 64 MFMAs (one GEMM stage), 4 = s_barrier per 64 MFMAs,
// 8 = LDS reads use the kernels' 32-byte-record pattern (2-way bank conflict), 16 = DMA streams a 1 GiB region (HBM, not L2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GP(p) ((const __attribute__((address_space(1))) void*)(p))
#define LP(p) ((__attribute__((address_space(3))) void*)(p))
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += 256) lds[i] = src[i];
  __syncthreads();
  f32x4 af[2], bf[2];
  af[0] = *(f32x4*)(lds + lane * 4); af[1] = *(f32x4*)(lds + 256 + lane * 4);
  bf[0] = *(f32x4*)(lds + 512 + lane * 4); bf[1] = *(f32x4*)(lds + 768 + lane * 4);
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  const float* g = src + (size_t)blockIdx.x * 8192;
  for (int it = 0; it < iters; ++it) {
    const int s = it & 1;
    f32x4 stg[8];
    if (MODE & 128) {  // register staging: plain 16-byte loads now, ds_write after the MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i)
        stg[i] = *(const f32x4*)((MODE & 16) ? (src + ((size_t)blockIdx.x * 1048576 + (size_t)(it & 127) * 8192 + ((i * 4 + wave) * 256 + lane * 4)))
                                             : (g + ((size_t)(it & 63) * 8192 + ((i * 4 + wave) * 256 + lane * 4)) % (1 << 22)));
    } else
    if ((MODE & 2) && (!(MODE & 32) || (it & 1)) && (!(MODE & 64) || (it & 3) == 0)) {  // 32: half the DMA volume, 64: a quarter
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds(GP((MODE & 16) ? (src + ((size_t)blockIdx.x * 1048576 + (size_t)(it & 127) * 8192 + ((i * 4 + wave) * 256 + lane * 4)))
                                                        : (g + ((size_t)(it & 63) * 8192 + ((i * 4 + wave) * 256 + lane * 4)) % (1 << 22))),
                                         LP(lds + (s ^ 1) * 8192 + (i * 4 + wave) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE & 1) {
        const float* L = lds + s * 8192 + kk * 1024 + ((MODE & 8) ? ((lane & 31) * 8 + (lane >> 5) * 4) : lane * 4);
        af[0] = *(const f32x4*)(L); af[1] = *(const f32x4*)(L + 256);
        bf[0] = *(const f32x4*)(L + 4096); bf[1] = *(const f32x4*)(L + 4096 + 256);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[0][j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][j], bf[1][j], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[0][j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][j], bf[1][j], acc[3], 0, 0, 0);
      }
    }
    if (MODE & 128) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f32x4*)(lds + (s ^ 1) * 8192 + (i * 4 + wave) * 256 + lane * 4) = stg[i];
    }
    if (MODE & 4) __syncthreads();
  }
  float sum = 0; for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
  out[blockIdx.x * 256 + tid] = sum;
}
template <int MODE> void run(const float* src, float* out) {
  const int iters = 4000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 65536, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 256.0 * 4 * iters * 64 * 2.0 * 32 * 32 * 2;
    if (rep) printf("mode(%d) ldsread=%d dma=%d barrier=%d conflict=%d hbm=%d : %.2f ms  %.1f TFLOP/s\n", MODE, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, (MODE >> 3) & 1, (MODE >> 4) & 1, ms, fl / ms / 1e9);
  }
}
int main() {
  float *src, *out; size_t n = (size_t)256 * 1048576 + (1 << 21);
  hipMalloc(&src, n * 4 + (256 * 8192 * 4)); hipMalloc(&out, 256 * 1024 * 4);
  std::vector<float> h(n + 256 * 8192); for (auto& v : h) v = rand() / (float)RAND_MAX * 2.f - 1.f;
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(src, out); run<0>(src, out); run<1>(src, out); run<9>(src, out); run<7>(src, out); run<15>(src, out); run<23>(src, out); run<31>(src, out); run<7 + 32>(src, out); run<7 + 64>(src, out); run<23 + 32>(src, out); run<23 + 64>(src, out); run<5 + 128>(src, out); run<21 + 128>(src, out);
}
