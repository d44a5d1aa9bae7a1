// VALU issue-rate probe (gfx950): cycles per wave-instruction for the ops of the attention softmax, with 1..4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_IT 256
#define CH 8
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, long* cyc, float seed) {
  float a[CH];
  f2 p[CH];
#pragma unroll
  for (int i = 0; i < CH; i++) { a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = (f2){a[i], a[i] * 0.5f}; }
  __syncthreads();
  long t0 = clock64();
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (OP == 1) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
      if (OP == 2) p[i] = __builtin_elementwise_fma(p[i], (f2){0.999f, 0.999f}, (f2){0.001f, 0.001f});
      if (OP == 3) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) % CH]), a[(i + 2) % CH]);
      if (OP == 4) { auto h = __builtin_amdgcn_cvt_pkrtz(a[i], a[(i + 1) % CH]); a[i] += (float)h[0]; }
      if (OP == 5) p[i] = p[i] * (f2){0.999f, 0.999f};
      if (OP == 6) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 7) p[i] = p[i] + (f2){0.001f, 0.002f};
    }
  }
  __syncthreads();  // every wave of the workgroup done (the scheduler favours the oldest wave)
  long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, int insts_per_iter_elem) {
  float* out; long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
  for (int waves = 4; waves <= 16; waves *= 2) {  // waves per workgroup on one CU -> waves/SIMD = waves/4
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 0.5f);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 0.5f);
    hipDeviceSynchronize();
    long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; i++) avg += h[i]; avg /= 256;
    double per = avg / (double)(N_IT * CH * insts_per_iter_elem) / (waves / 4.0);
    printf("%-22s waves/SIMD %d: %8.0f cycles total, %.2f cycles per wave-instruction per SIMD\n", name, waves / 4, avg, per);
  }
}
int main() {
  run<0>("v_exp_f32", 1);
  run<6>("v_rcp_f32", 1);
  run<1>("v_fma_f32", 1);
  run<2>("v_pk_fma_f32", 1);
  run<5>("v_pk_mul_f32", 1);
  run<7>("v_pk_add_f32", 1);
  run<3>("v_max3_f32", 1);
  run<4>("cvt_pkrtz + cvt + add", 3);
  return 0;
}
