// probe: what the CU's vector-memory path delivers beside the matrix pipe.
//   loads: buffer_load_dwordx4 to registers (VGPR), global_load_lds_dwordx4 (LDSDMA: straight into LDS, no register write) or
//   ds_read_b128 (LDS -> registers); one 1-KiB contiguous piece per wave instruction, 12 in flight per wave, 8 waves per CU;
//   from L1 (all waves re-read one 1-KiB window), L2 (2 MiB), all L2s (16 MiB), Infinity Cache (64 MiB), HBM (2 GiB);
//   alone and with 8 independent bf16 MFMAs (random operands) between every 3 loads — the B-direct convolution's k-step.
// Every case runs >= 0.3 s while the host samples the GPU's sclk / package power from sysfs hwmon.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <glob.h>
#include <string>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// KIND: 0 no loads, 1 VGPR loads, 2 LDS-DMA loads, 3 ds_read_b128
template <int MF, int KIND, int SCH = 0>
__global__ __launch_bounds__(256, 2) void k(const u32x4* __restrict__ src, unsigned mask, int iters, float* out, const bf16x8* __restrict__ ops) {
  __shared__ u32x4 lds[4][12][64];  // per wave: 12 pieces of 1 KiB (48 KiB per block)
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * 4 + wv;
  constexpr int R = 12;
  u32x4 ring[R];
  for (int r = 0; r < R; ++r) { ring[r] = u32x4{0, 0, 0, 0}; lds[wv][r][lane] = u32x4{lane, wv, (unsigned)r, 1u}; }
  bf16x8 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = ops[threadIdx.x * 6 + i];
  for (int i = 0; i < 2; ++i) b[i] = ops[threadIdx.x * 6 + 4 + i];
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  unsigned rec = w * 977u * 64u;
  unsigned sum = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    auto load1 = [&](int slot) {
      if (KIND == 1) {
        asm volatile("" :: "v"(ring[slot]));  // the previous load into this slot has to have landed
        ring[slot] = src[((rec + lane) & mask)];
      } else if (KIND == 2) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((rec + lane) & mask)),
                                         (__attribute__((address_space(3))) void*)(&lds[wv][slot][0]), 16, 0, 0);
      } else if (KIND == 3) {
        asm volatile("" :: "v"(ring[slot]));
        ring[slot] = lds[wv][(slot + it) % 12][lane];
      }
      rec += 64;
    };
    auto mfma1 = [&](int t) {
      __builtin_amdgcn_sched_barrier(0);
      acc[t & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t & 7) >> 1], b[t & 1], acc[t & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (SCH == 3) {
#pragma unroll
      for (int j = 0; j < R; ++j) load1(j);
      if (KIND == 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4 * MF; ++t) mfma1(t);
    } else {
#pragma unroll
      for (int g = 0; g < R / 3; ++g) {
        if (SCH == 1) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            load1(g * 3 + j);
#pragma unroll
            for (int t = 0; t < (j < 2 ? 3 : 2) && MF; ++t) mfma1(j * 3 + t);
          }
        } else if (SCH == 4) {  // 2 loads per 8 MFMAs, dealt (a 256 x 256 LDS-shared tile's 7.6 KiB per MFLOP)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            load1(g * 3 + j);
#pragma unroll
            for (int t = 0; t < 4 && MF; ++t) mfma1(j * 4 + t);
          }
        } else if (SCH == 5) {  // 1 load per 8 MFMAs
          load1(g * 3);
#pragma unroll
          for (int t = 0; t < MF; ++t) mfma1(t);
        } else {
#pragma unroll
          for (int j = 0; j < 3; ++j) load1(g * 3 + j);
          if (SCH == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int t = 0; t < MF; ++t) mfma1(t);
          if (SCH == 2) __builtin_amdgcn_s_setprio(0);
        }
        if (KIND == 2 && g == R / 3 - 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // keep <= 9 DMA pieces in flight
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int r = 0; r < R; ++r) sum += ring[r][0] + lds[wv][r][lane][0];
  float s = (float)sum; for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static std::string g_freq, g_power;
static double read_num(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return 0; double v = 0; if (fscanf(f, "%lf", &v) != 1) v = 0; fclose(f); return v; }
template <int MF, int KIND, int SCH = 0> void run(const u32x4* src, size_t window_bytes, float* out, const bf16x8* ops, const char* what) {
  const int blocks = 512;
  const unsigned mask = (unsigned)(window_bytes / 16 - 1);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MF, KIND, SCH>), dim3(blocks), dim3(256), 0, 0, src, mask, iters, out, ops);
    (void)hipEventRecord(e1);
    double fsum = 0, psum = 0; int n = 0;
    while (hipEventQuery(e1) == hipErrorNotReady) { if (rep) { fsum += read_num(g_freq); psum += read_num(g_power); ++n; } }
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) { iters = (int)(iters * 400.0 / ms) + 1; continue; }
    const double bytes = KIND ? (double)blocks * 4 * iters * (SCH == 4 ? 8 : SCH == 5 ? 4 : 12) * 1024.0 : 0.0;
    const double fl = (double)blocks * 4 * iters * 4 * MF * 2.0 * 32 * 32 * 16;
    const char* kn[] = {"no loads", "-> VGPR", "-> LDS (DMA)", "ds_read_b128"};
    printf("%-24s %-13s sched %d MFMAs %d: %7.1f GB/s per CU %6.2f TB/s  MFMA %7.1f TFLOP/s   sclk %4.0f MHz  %4.0f W   -> %5.1f B/clk/CU, matrix pipe %3.0f %% busy\n", what, kn[KIND], SCH, MF,
           bytes / ms / 1e6 / 256, bytes / ms / 1e9, fl / ms / 1e9, n ? fsum / n / 1e6 : 0.0, n ? psum / n / 1e6 : 0.0,
           n && fsum > 0 ? bytes / ms / 1e6 / 256 / (fsum / n / 1e9) * 1e-0 / 1.0 * 1e0 / 1e0 * 1.0 / 1.0 * 1e-0 : 0.0,
           n && fsum > 0 ? 100.0 * fl / ms / 1e9 / (2500.0 * (fsum / n / 1e6) / 2400.0) : 0.0);
  }
}
int main() {
  glob_t g;
  if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", 0, nullptr, &g) == 0 && g.gl_pathc) {
    g_freq = g.gl_pathv[0];
    g_power = g_freq.substr(0, g_freq.rfind('/')) + "/power1_input";
    if (read_num(g_power) == 0) g_power = g_freq.substr(0, g_freq.rfind('/')) + "/power1_average";
  }
  const size_t big = (size_t)2 << 30;
  u32x4* src; float* out; bf16x8* ops;
  (void)hipMalloc(&src, big); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&ops, 256 * 6 * 16);
  (void)hipMemset(src, 1, big);
  std::vector<unsigned short> h(256 * 6 * 8);
  for (auto& v : h) v = (unsigned short)(0x3c00u + (rand() & 0x3ff) + ((rand() & 1) << 15));
  (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  run<8, 0>(src, 1024, out, ops, "-");
  run<0, 3>(src, 1024, out, ops, "LDS");
  run<8, 3>(src, 1024, out, ops, "LDS");
  const struct { size_t w; const char* n; } ws[] = {{1024, "L1 (one 1-KiB window)"}, {(size_t)2 << 20, "L2 (2 MiB)"}, {(size_t)16 << 20, "L2 of 8 XCDs (16 MiB)"},
                                                   {(size_t)64 << 20, "Infinity Cache (64 MiB)"}, {big, "HBM (2 GiB)"}};
  for (auto& x : ws) {
    run<0, 1>(src, x.w, out, ops, x.n); run<8, 1>(src, x.w, out, ops, x.n);
    run<0, 2>(src, x.w, out, ops, x.n); run<8, 2>(src, x.w, out, ops, x.n);
  }
  // schedules (L2-resident window): 0 = 3 loads then 8 MFMAs, 1 = a load every 2-3 MFMAs, 2 = schedule 0 with s_setprio 1 over the MFMAs, 3 = 12 loads then 32 MFMAs
  run<8, 1, 1>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 1, 2>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 1, 3>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)");
  run<8, 2, 1>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 2, 2>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 2, 3>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)");
  // fewer bytes per FLOP, dealt: 4 = 2 loads per 8 MFMAs (7.6 KiB / MFLOP), 5 = 1 load per 8 MFMAs (3.8)
  run<8, 1, 4>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 2, 4>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)");
  run<8, 1, 5>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)"); run<8, 2, 5>(src, (size_t)2 << 20, out, ops, "L2 (2 MiB)");
  run<8, 1, 1>(src, (size_t)16 << 20, out, ops, "L2 of 8 XCDs (16 MiB)"); run<8, 1, 4>(src, (size_t)16 << 20, out, ops, "L2 of 8 XCDs (16 MiB)");
  run<8, 1, 1>(src, (size_t)64 << 20, out, ops, "Infinity Cache (64 MiB)"); run<8, 1, 4>(src, (size_t)64 << 20, out, ops, "Infinity Cache (64 MiB)");
}
