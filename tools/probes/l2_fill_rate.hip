// L2 -> CU operand-stream probe (gfx950): how many bytes per second can one workgroup per CU pull from an L2-resident buffer
//   V0  global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction), DEPTH instructions in flight per wave, counted vmcnt
//   V1  global_load_dwordx4 -> VGPR -> ds_write_b128 (register staging)
//   V2  global_load_dwordx4 -> VGPR only (xor-accumulated)
// with the access shapes of the GEMM staging (8 rows x 128 B per wave-instruction at a row stride) or contiguous 1 KiB.
// Question it answers: is the ~10 TB/s "L2 -> LDS fill" plateau of the GEMM k-loops a property of LDS-DMA, of the
// LDS write port, or of the L2 -> CU path itself?
// build: hipcc --offload-arch=gfx950 -O3 -o l2_fill_rate l2_fill_rate.hip ; run: ./l2_fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// SW: lane -> source map inside a piece (the LDS image stays lane-linear, as LDS-DMA requires)
//   0  8 rows x 128 B, chunks in lane order            1  same rows, chunk index XOR (row & 7)  (the GEMMs' bank swizzle, BK = 64)
//   2  16 rows x 64 B, chunks in lane order (BK = 32)  3  16 rows x 64 B with gemm.hip's pair swizzle ((s & 7) ^ (p & 7))
//   4  8 rows x 128 B, chunk index rotated by the row
template <int V, int DEPTH, int SW = 0>
__global__ void fill(const char* __restrict__ src, size_t bytes_mask, long rowstride, int iters, uint4* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  // lane offset inside a piece: rowstride == 0 -> 1 KiB contiguous; else 8 rows x 128 B
  long lane_off = rowstride ? (long)(lane >> 3) * rowstride + (lane & 7) * 16 : (long)lane * 16;
  long piece = rowstride ? 8 * rowstride : 1024;  // address step between consecutive pieces of one wave
  if (SW == 1) lane_off = (long)(lane >> 3) * rowstride + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
  if (SW == 4) lane_off = (long)(lane >> 3) * rowstride + ((((lane & 7) + (lane >> 3)) & 7) * 16);
  if (SW == 2) { lane_off = (long)(lane >> 2) * rowstride + (lane & 3) * 16; piece = 16 * rowstride; }
  if (SW == 3) {  // gemm.hip DmaPlan MODE 0: slot s -> pair p = s >> 3, x = (s & 7) ^ (p & 7), row = 2 p + (x >> 2), chunk = x & 3
    const int pr = lane >> 3, x = (lane & 7) ^ (pr & 7);
    lane_off = (long)(2 * pr + (x >> 2)) * rowstride + (x & 3) * 16;
    piece = 16 * rowstride;
  }
  // start positions differ per workgroup and wave (CUs of a GEMM read different panels at the same time)
  size_t pos = ((size_t)blockIdx.x * 7919 * 4096 + (size_t)wave * piece * 17) & bytes_mask;
  char* my = smem + (size_t)wave * DEPTH * 1024;
  uint4 acc = {0, 0, 0, 0};
  if (V == 0) {
    for (int it = 0; it < iters; it++) {
      const char* p = src + ((pos + lane_off) & bytes_mask);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(my + (it % DEPTH) * 1024), 16, 0, 0);
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      pos = (pos + (size_t)piece * nw) & bytes_mask;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<uint4*>(my + lane * 16);
  } else {
    uint4 r[DEPTH], r2[DEPTH];
    auto load = [&](uint4* dst) {
#pragma unroll
      for (int j = 0; j < DEPTH; j++) {
        dst[j] = *reinterpret_cast<const uint4*>(src + ((pos + lane_off) & bytes_mask));
        pos = (pos + (size_t)piece * nw) & bytes_mask;
      }
    };
    auto sinkit = [&](uint4* v) {
#pragma unroll
      for (int j = 0; j < DEPTH; j++) {
        if (V == 1) *reinterpret_cast<uint4*>(my + j * 1024 + lane * 16) = v[j];
        else { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
      }
    };
    load(r);
    for (int it = 0; it < iters; it += 2 * DEPTH) {  // two register sets: the next batch is requested before this one is consumed
      load(r2);
      sinkit(r);
      load(r);
      sinkit(r2);
    }
    sinkit(r);
    if (V == 1) { __syncthreads(); acc = *reinterpret_cast<uint4*>(my + lane * 16); }
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x * blockDim.x + tid] = acc;  // never true in practice: keeps the loads alive
}

template <int V, int DEPTH, int SW = 0>
void run(const char* name, const char* src, size_t bytes, long rowstride, int waves, int wgs_per_cu, uint4* sink) {
  const int iters = 1024;
  const int grid = 256 * wgs_per_cu;
  const size_t lds = (size_t)waves * DEPTH * 1024;
  auto kern = fill<V, DEPTH, SW>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, src, bytes - 1, rowstride, iters, sink);
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, src, bytes - 1, rowstride, iters, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double total = (double)grid * waves * (iters + (V ? 0 : 0)) * 1024.0;
  const double tbs = total / (ms * 1e-3) / 1e12;
  printf("%-34s depth %2d waves/WG %2d WG/CU %d buf %6zu KiB stride %5ld : %7.1f us  %6.2f TB/s  %5.1f GB/s per CU  (%4.1f B/clk @2.4GHz)\n",
         name, DEPTH, waves, wgs_per_cu, bytes >> 10, rowstride, ms * 1e3, tbs, tbs * 1e3 / 256, tbs * 1e12 / 256 / 2.4e9);
}

int main(int argc, char** argv) {
  const size_t big = (size_t)1 << 30;
  char* src;
  uint4* sink;
  CHECK(hipMalloc(&src, big));
  CHECK(hipMemset(src, 1, big));
  CHECK(hipMalloc(&sink, 64 << 20));
  if (argc > 1) {  // lane -> source maps of the GEMM staging (L2-resident 2 MiB buffer, 8 waves, depth 4)
    const size_t b = (size_t)2 << 20;
    for (long stride : {1024L, 6144L}) {
      run<0, 4, 0>("LDS-DMA 8 rows x 128 B, lane order", src, b, stride, 8, 1, sink);
      run<0, 4, 1>("LDS-DMA 8 rows x 128 B, XOR swizzle", src, b, stride, 8, 1, sink);
      run<0, 4, 4>("LDS-DMA 8 rows x 128 B, rotation", src, b, stride, 8, 1, sink);
      run<0, 4, 2>("LDS-DMA 16 rows x 64 B, lane order", src, b, stride, 8, 1, sink);
      run<0, 4, 3>("LDS-DMA 16 rows x 64 B, pair swizzle", src, b, stride, 8, 1, sink);
      run<2, 4, 0>("load x4 -> VGPR 8 rows x 128 B", src, b, stride, 8, 1, sink);
      run<2, 4, 1>("load x4 -> VGPR, XOR swizzle", src, b, stride, 8, 1, sink);
      run<2, 4, 3>("load x4 -> VGPR 16 x 64 B pair swizzle", src, b, stride, 8, 1, sink);
    }
    return 0;
  }
  const size_t sizes[3] = {(size_t)2 << 20, (size_t)64 << 20, big};
  for (int s = 0; s < 3; s++) {
    const size_t b = sizes[s];
    for (long stride : {0L, 1024L}) {
      run<0, 4>("LDS-DMA x4", src, b, stride, 8, 1, sink);
      run<0, 8>("LDS-DMA x4", src, b, stride, 8, 1, sink);
      run<0, 16>("LDS-DMA x4", src, b, stride, 8, 1, sink);
      run<0, 8>("LDS-DMA x4", src, b, stride, 4, 2, sink);
      run<0, 8>("LDS-DMA x4", src, b, stride, 16, 1, sink);
      run<1, 4>("load x4 -> VGPR -> ds_write_b128", src, b, stride, 8, 1, sink);
      run<1, 8>("load x4 -> VGPR -> ds_write_b128", src, b, stride, 8, 1, sink);
      run<1, 4>("load x4 -> VGPR -> ds_write_b128", src, b, stride, 16, 1, sink);
      run<1, 4>("load x4 -> VGPR -> ds_write_b128", src, b, stride, 8, 2, sink);
      run<2, 4>("load x4 -> VGPR", src, b, stride, 8, 1, sink);
      run<2, 8>("load x4 -> VGPR", src, b, stride, 8, 1, sink);
      run<2, 8>("load x4 -> VGPR", src, b, stride, 16, 1, sink);
      run<2, 8>("load x4 -> VGPR", src, b, stride, 8, 2, sink);
    }
  }
  return 0;
}
