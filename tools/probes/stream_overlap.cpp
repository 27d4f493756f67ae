// probe: do two kernels on two streams of one process share the GPU?  Each kernel = `blocks` blocks of 256 threads spinning for ~1 ms with 246-VGPR-like
// occupancy irrelevant (tiny kernel): 384 blocks on 256 CUs.  Prints the wall time of A alone, and of A and B launched on two streams
// (normal + normal, normal + lowest priority), each after a fork event like the tower lanes use.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long cycles, float* out) {
  const long long t0 = clock64();
  float x = threadIdx.x;
  while (clock64() - t0 < cycles) x = x * 1.0001f + 0.5f;
  if (x == 123.456f) out[0] = x;
}
static float run(hipStream_t a, hipStream_t b, int blocks, long long cyc, float* out, bool two) {
  hipEvent_t e0, e1, fork, join;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
  hipEventRecord(e0, a);
  if (two) { hipEventRecord(fork, a); hipStreamWaitEvent(b, fork, 0); }
  for (int i = 0; i < 10; ++i) {
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, cyc, out);
    if (two) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, b, cyc, out);
  }
  if (two) { hipEventRecord(join, b); hipStreamWaitEvent(a, join, 0); }
  hipEventRecord(e1, a); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipStream_t a, b, c; int lo, hi;
  hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking); hipStreamCreateWithPriority(&c, hipStreamNonBlocking, lo);
  const long long cyc = 100000;  // ~1 ms at 100 MHz clock64
  for (int blocks : {128, 384, 768}) {
    run(a, b, blocks, cyc, out, false);
    printf("blocks %4d: one stream x10 %7.2f ms | two streams (normal + normal) x10 each %7.2f ms | (normal + lowest priority) %7.2f ms | default stream + normal %7.2f ms\n", blocks,
           run(a, b, blocks, cyc, out, false), run(a, b, blocks, cyc, out, true), run(a, c, blocks, cyc, out, true), run(0, b, blocks, cyc, out, true));
  }
}
