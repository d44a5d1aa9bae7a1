// Probe (round 5): what the matrix pipe of this chip SUSTAINS, measured so that clock ramp and launch edges cannot
// colour it.  Every SIMD issues nothing but v_mfma_f32_32x32x16_f16 (one dependent accumulator chain per wave, one wave per SIMD):
// n = 2 000 000 MFMAs per wave per launch (>= 40 ms), >= 0.5 s of the same kernel as warm-up, then 5 timed launches.  Reported per
// operand set (all-zero / uniform random in [-1, 1)): chip TFLOP/s from hipEvents, and the shader clock seen INSIDE the kernel
// (s_memtime ticks over s_memrealtime's constant 100 MHz).  Run under tools/power_probe.sh for the package power next to it.
// The 2.5 PF denominator of every `frac` in this repository is MI355X_MICROARCH.md's (2495 TF measured); whatever this prints is a
// second column, never a replacement.
// Build: hipcc --offload-arch=gfx950 -O2 -w tools/probes/mfma_long.hip -o tools/probes/mfma_long
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ ab, float* out, unsigned long long* clk, int n) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = ab[(threadIdx.x * 8 + i) & 4095]; b[i] = ab[4096 + ((threadIdx.x * 8 + i) & 4095)]; }
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < n; it += 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2000000, blocks = 256, threads = 256;
  _Float16* ab; float* out; unsigned long long* clk;
  hipMalloc(&ab, 8192 * sizeof(_Float16)); hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&clk, 16 * blocks);
  std::vector<_Float16> h(8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    srand(1);
    for (auto& x : h) x = (_Float16)(mode ? (rand() / (float)RAND_MAX * 2.f - 1.f) : 0.f);
    hipMemcpy(ab, h.data(), 8192 * sizeof(_Float16), hipMemcpyHostToDevice);
    const double flop = (double)blocks * (threads / 64) * n * 32768.0;  // 2 * 32 * 32 * 16 per instruction
    float warm_ms = 0.f;
    int warm = 0;
    while (warm_ms < 500.f) {  // >= 0.5 s of the same kernel before anything is timed
      hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, ab, out, clk, n); hipEventRecord(e1, 0);
      hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); warm_ms += ms; warm++;
    }
    double best = 0, sum = 0, ghz = 0;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, ab, out, clk, n); hipEventRecord(e1, 0);
      hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1);
      const double tf = flop / (ms * 1e-3) * 1e-12; sum += tf; if (tf > best) best = tf;
      std::vector<unsigned long long> c(2 * blocks);
      hipMemcpy(c.data(), clk, 16 * blocks, hipMemcpyDeviceToHost);
      double g = 0; for (int b = 0; b < blocks; b++) g += (double)c[2 * b] / (double)c[2 * b + 1] * 0.1; ghz = g / blocks;
    }
    printf("%-7s operands: %d MFMAs per wave per launch, %d warm-up launches (%.0f ms): mean %.0f TFLOP/s, best %.0f; %.2f ns per MFMA per "
           "SIMD; s_memtime / s_memrealtime inside the kernel = %.3f GHz (if s_memtime ticks at the shader clock; 0.100 = it does not)\n",
           mode ? "random" : "zero", n, warm, warm_ms, sum / 5, best, 32768.0 * 1024.0 / (sum / 5 * 1e12) * 1e9, ghz);
  }
  return 0;
}
