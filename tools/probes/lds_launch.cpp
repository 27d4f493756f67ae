// probe: launch cost of a trivially-exiting kernel vs dynamic LDS size and block count
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const int* flag, int* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long l[];
  if (flag[blockIdx.x] != 1) return;
  l[threadIdx.x] = threadIdx.x; __syncthreads(); out[blockIdx.x] = (int)l[255 - threadIdx.x];
}
int main() {
  int *flag, *out; hipMalloc(&flag, 4096); hipMalloc(&out, 4096); hipMemset(flag, 0, 4096);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {20, 256}) for (int kb : {0, 16, 64, 96, 128, 144, 160}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), kb * 1024, 0, flag, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), kb * 1024, 0, flag, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("blocks=%d lds=%dKB  %.1f us/launch\n", blocks, kb, ms / 20 * 1e3);
  }
}
