// Epilogue store-pattern probe (gfx950): 256 workgroups x 8 waves, every wave stores a 64-row x 128-column 16-bit block of a
// row-major [M][ld] matrix (a 256 x 256 GEMM tile per workgroup, gemm3's decomposition) with different lane -> address maps:
//   A  8 B / lane, lanes m = l&15 -> 16 rows, g = l>>4 -> 4 consecutive 8-B pieces: 16 rows x 32 B per instruction (gemm_epi3 today)
//   B  16 B / lane, 16 rows x 64 B per instruction (what a v_permlane16_swap pairing of column quads gives)
//   C  16 B / lane, 4 rows x 256 B per instruction (LDS-transposed: 16 lanes cover a wave's 256-B row segment)
//   D  16 B / lane, 8 rows x 128 B per instruction (head-major q/k/v: 64 elements per (token, head))
//   E  as C with 8 B / lane (2 rows x 256 B... per instruction 512 B): separates "bytes per lane" from "lines per instruction"
// Reports us per pass and GB/s per CU.  The question: is the register epilogue's ~15 GB/s per CU (tools/native/gemm_trace.cpp:
// 128 KiB per workgroup in 8.6 us) a property of the 32-byte runs?
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int P>
__global__ __launch_bounds__(512) void st(unsigned short* __restrict__ C, long ld, int tiles_n, unsigned seed) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int wr = wave >> 1, wc = wave & 1;
  // wave block: rows tm*256 + (half)*128 + wr*32 + 0..31 for half = 0,1 ; columns tn*256 + wc*128 + 0..127
  unsigned short* base = C + ((long)tm * 256 + wr * 32) * ld + tn * 256 + wc * 128;
  const uint2 v2 = make_uint2(seed + tid, seed * 3 + tid);
  const uint4 v4 = make_uint4(seed + tid, seed * 3 + tid, seed * 5 + tid, seed * 7 + tid);
  const int m = lane & 15, g = lane >> 4;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    unsigned short* hb = base + (long)half * 128 * ld;
    if (P == 0) {  // A: i = 0,1 (16-row groups), j = 0..7 (16-column blocks): 8 B at [16 i + m][16 j + 4 g]
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) *reinterpret_cast<uint2*>(hb + (long)(16 * i + m) * ld + 16 * j + 4 * g) = v2;
    } else if (P == 1) {  // B: 16 B at [16 i + m][32 jp + 8 g], jp = 0..3
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jp = 0; jp < 4; jp++) *reinterpret_cast<uint4*>(hb + (long)(16 * i + m) * ld + 32 * jp + 8 * g) = v4;
    } else if (P == 2) {  // C: 16 B at [4 it + g][8 m], it = 0..7
#pragma unroll
      for (int it = 0; it < 8; it++) *reinterpret_cast<uint4*>(hb + (long)(4 * it + g) * ld + 8 * m) = v4;
    } else if (P == 3) {  // D: two 128-B segments per row at a large distance (heads): 8 rows x 128 B per instruction, 2 per row pair
#pragma unroll
      for (int it = 0; it < 4; it++)
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
          *reinterpret_cast<uint4*>(hb + (long)(8 * it + (lane >> 3)) * ld + hh * 64 + 8 * (lane & 7)) = v4;
    } else {  // E: 8 B per lane, 2 rows x 256 B per instruction
#pragma unroll
      for (int it = 0; it < 16; it++) *reinterpret_cast<uint2*>(hb + (long)(2 * it + (lane >> 5)) * ld + 4 * (lane & 31)) = v2;
    }
  }
}

template <int P>
void run(const char* name, unsigned short* C, long ld, int tiles_n, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(st<P>, dim3(grid), dim3(512), 0, 0, C, ld, tiles_n, (unsigned)i);
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(st<P>, dim3(grid), dim3(512), 0, 0, C, ld, tiles_n, (unsigned)i);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)grid * 256 * 256 * 2;
  printf("%-46s grid %4d: %6.2f us per launch, %6.2f TB/s, %6.1f GB/s per workgroup-CU\n", name, grid, us, bytes / us * 1e-6, bytes / grid / us * 1e-3);
}

int main() {
  const long ld = 3072;
  const int tiles_n = 12;
  unsigned short* C;
  CHECK(hipMalloc(&C, (size_t)33 * 256 * ld * 2 + (1 << 20)));
  for (int grid : {256, 128, 384}) {
    run<0>("A  8 B/lane, 16 rows x 32 B (today)", C, ld, tiles_n, grid);
    run<1>("B 16 B/lane, 16 rows x 64 B (permlane pairs)", C, ld, tiles_n, grid);
    run<2>("C 16 B/lane, 4 rows x 256 B (LDS transposed)", C, ld, tiles_n, grid);
    run<3>("D 16 B/lane, 8 rows x 128 B", C, ld, tiles_n, grid);
    run<4>("E  8 B/lane, 2 rows x 256 B", C, ld, tiles_n, grid);
  }
  return 0;
}
