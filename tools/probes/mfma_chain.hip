// Probe (round 3): how long is the dependent-accumulate chain of v_mfma_f32_32x32x16_{f16,bf16} on gfx950, and how many waves /
// independent chains does a SIMD need to keep its matrix pipe busy?  One workgroup per CU; W waves per SIMD (argv[1]: 1..4) each issue
// N MFMAs as C independent accumulator chains (C = 1, 2, 4) and the shader clock (s_memtime) is read around the loop.
// Output: chip TFLOP/s and ns per MFMA per SIMD from hipEvents around the launch.
// Also: the same loop with a ds_read_b128 + s_waitcnt lgkmcnt(0) in front of every group of 4 MFMAs (the attention kernels' phase
// pattern: fragment read -> wait -> dependent MFMAs), which prices the exposed LDS latency per phase.
// Build: hipcc --offload-arch=gfx950 -O2 -w tools/probes/mfma_chain.hip -o tools/probes/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int CH, bool LDS>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int n) {
  __shared__ __attribute__((aligned(16))) char sm[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(sm)[i] = 0.001f * i;
  __syncthreads();
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (lane + i)); b[i] = (_Float16)(0.02f * (lane - i)); }
  f32x16 acc[CH];
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
  const unsigned addr = (unsigned)(size_t)sm + lane * 16;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; it += 4 * CH) {
    if (LDS) {
      f32x4 v;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      a[0] = (_Float16)v[0];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 16; i++) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double g_tflops = 0;
template <int CH, bool LDS>
static double run(int waves_per_simd, int n) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CH, LDS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<CH, LDS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  g_tflops = (double)blocks * (threads / 64) * n * 32768.0 /* 2*32*32*16 FLOP per v_mfma_f32_32x32x16 */ / (ms * 1e-3) * 1e-12;
  std::vector<unsigned long long> h(blocks * 16);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost);
  double tot = 0; int cnt = 0;
  for (int b = 0; b < blocks; b++) for (int w = 0; w < threads / 64; w++) { tot += (double)h[b * 16 + w]; cnt++; }
  hipFree(out); hipFree(cyc);
  // s_memtime counts at the constant 100 MHz reference on this part; report wave-time per MFMA of ONE wave in reference ticks * 1000
  return tot / cnt / n * waves_per_simd;  // ticks per MFMA per wave, times waves sharing the SIMD = ticks per MFMA slot of the SIMD... see main
}

int main(int argc, char** argv) {
  const int n = 4096;
  printf("v_mfma_f32_32x32x16_f16, every SIMD of the chip busy; W waves per SIMD, C accumulator chains per wave\n");
  printf("chip TFLOP/s and ns per MFMA per SIMD (hipEvent over the launch)\n");
  for (int w = 1; w <= 4; w++) {
    double t[6];
    run<1, false>(w, n); t[0] = g_tflops; run<2, false>(w, n); t[1] = g_tflops; run<4, false>(w, n); t[2] = g_tflops;
    run<1, true>(w, n); t[3] = g_tflops; run<2, true>(w, n); t[4] = g_tflops; run<4, true>(w, n); t[5] = g_tflops;
    auto ns = [](double tf) { return 32768.0 * 1024.0 / (tf * 1e12) * 1e9; };
    printf("W=%d  MFMA only: C=1 %.0f (%.1f ns)  C=2 %.0f  C=4 %.0f (%.1f ns) | ds_read_b128 + lgkmcnt(0) before every 4*C MFMAs: C=1 %.0f (%.1f ns)  C=2 %.0f  C=4 %.0f\n",
           w, t[0], ns(t[0]), t[1], t[2], ns(t[2]), t[3], ns(t[3]), t[4], t[5]);
  }
  return 0;
}
