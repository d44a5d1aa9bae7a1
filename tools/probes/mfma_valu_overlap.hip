// Can VALU work hide under MFMA work on one SIMD (gfx950)?  All instructions are inline asm so the mix is exact.
//   MODE 0: MFMA only   MODE 1: VALU only   MODE 2: same wave, 2 MFMA then 2R VALU, repeated
//   MODE 3: waves split by role (wave-uniform branch): waves 0-3 of the workgroup MFMA-only, waves 4-7 VALU-only
// VOP: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_exp_f32, 3 v_max3_f32, 4 v_cvt_pkrtz_f16_f32
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define N_IT 128
template <int VOP>
__device__ __forceinline__ void valu(f2& p, float& x, float c1, float c2) {
  if (VOP == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"((f2){c1, c1}), "v"((f2){c2, c2}));
  if (VOP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
  if (VOP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (VOP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
  if (VOP == 4) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
}
template <int MODE, int R, int VOP>
__global__ __launch_bounds__(512) void k(float* out, long* cyc, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
  f32x16 acc0 = {0}, acc1 = {0};
  f2 p[8];
  float x[8];
  for (int i = 0; i < 8; i++) { p[i] = (f2){seed + i, seed * 0.5f + i}; x[i] = seed + i; }
  const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
  const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
  __syncthreads();
  long t0 = clock64();
  for (int it = 0; it < N_IT; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (do_mfma) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
      }
      if (do_valu) {
#pragma unroll
        for (int r = 0; r < 2 * R; r++) valu<VOP>(p[r % 8], x[r % 8], 0.999f, 0.001f);
      }
    }
  }
  __syncthreads();
  long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; i++) s += p[i].x + p[i].y + x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int R, int VOP>
double run() {
  float* out; long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
  hipLaunchKernelGGL((k<MODE, R, VOP>), dim3(256), dim3(512), 0, 0, out, cyc, 0.5f);
  hipLaunchKernelGGL((k<MODE, R, VOP>), dim3(256), dim3(512), 0, 0, out, cyc, 0.5f);
  hipDeviceSynchronize();
  long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; i++) avg += h[i]; avg /= 256;
  hipFree(out); hipFree(cyc);
  return avg;
}
template <int VOP>
void suite(const char* name) {
  const double m = run<0, 8, VOP>(), v = run<1, 8, VOP>(), same = run<2, 8, VOP>(), split = run<3, 8, VOP>();
  printf("%-20s 2 waves/SIMD, per wave 1024 MFMA 32x32x16 and/or 8192 VALU:  MFMA-only %7.0f  VALU-only %7.0f  same-wave mix %7.0f "
         "(sum %7.0f, max %7.0f)   waves split by role (1 MFMA wave + 1 VALU wave per SIMD) %7.0f\n", name, m, v, same, m + v,
         m > v ? m : v, split);
}
int main() {
  suite<0>("v_pk_fma_f32");
  suite<1>("v_fma_f32");
  suite<2>("v_exp_f32");
  suite<3>("v_max3_f32");
  suite<4>("v_cvt_pkrtz_f16_f32");
  return 0;
}
