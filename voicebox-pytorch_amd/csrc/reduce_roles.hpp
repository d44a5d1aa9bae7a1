// Bodies of the two "several reductions in one launch" kernels (vbx_splitk_reduce_multi: gemm.hip, vbx_multi_reduce: ops.hip) as
// device functions, so that vbx_layer_reduce (ops.hip) can run both job tables of a layer's backward in ONE launch.
#pragma once
#include "common.hpp"

// split-K slab reduction of four columns of one row: i4 = flat index (multiple of 4) into the [M, N] result of job jb.
// Returns the sum of squares of the values it STORED (0 for dropped rows / columns): the gradient-norm term of this thread.
VBX_DEV float skr_role(const vbx_skr_job& jb, long i4) {
  const long total = (long)jb.M * jb.N;
  if (i4 >= total) return 0.f;
  const int r = (int)(i4 / jb.N), c = (int)(i4 - (long)r * jb.N);
  int dr = r;
  if (jb.rowmap == 1) dr = geglu_row_unmap(r, jb.F);
  if (dr < 0 || dr >= jb.dst_rows) return 0.f;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* sp = jb.slabs + i4;
  int k = 0;
  for (; k + 4 <= jb.splits; k += 4) {  // four slabs requested before the first is consumed (same summation order)
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(sp + (long)(k + u) * total);
#pragma unroll
    for (int u = 0; u < 4; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; k < jb.splits; k++) {
    const float4 v = *reinterpret_cast<const float4*>(sp + (long)k * total);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = jb.dst + (long)dr * jb.dst_ld + c;
  float sq = 0.f;
  if (c + 3 < jb.dst_cols && (jb.dst_ld & 3) == 0) {
    *reinterpret_cast<float4*>(o) = s;
    sq = (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
  } else {
    const float t[4] = {s.x, s.y, s.z, s.w};
    for (int e = 0; e < 4; e++)
      if (c + e < jb.dst_cols) { o[e] = t[e]; sq += t[e] * t[e]; }
  }
  return sq;
}

// Block sum in a FIXED order (wave butterfly, then the wave partials in index order): the same value on every run.  Every thread of
// the block must call it; the result is valid in thread 0.  wsum: >= blockDim.x / 64 floats of LDS.
VBX_DEV float block_sum_fixed(float v, float* wsum) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) wsum[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; w++) t += wsum[w];
  return t;
}

// column reduction of job jb by a 1024-thread block (64 columns x 16 row lanes); local = block index inside the job
VBX_DEV void mr_role(const vbx_mr_job& jb, int local, float (*red)[64]) {
  const int cblocks = (jb.cols + 63) >> 6;
  const int b = local / cblocks, cb = local - b * cblocks;
  const int il = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = cb * 64 + il;
  float s = 0.f;
  if (c < jb.cols) {
    const float* p = jb.src + (long)b * jb.src_bstride + c;
    int r = rl;
    for (; r + 48 < jb.rows; r += 64) {  // four rows requested before the first is consumed (same summation order)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = p[(long)(r + 16 * u) * jb.row_stride];
#pragma unroll
      for (int u = 0; u < 4; u++) s += v[u];
    }
    for (; r < jb.rows; r += 16) s += p[(long)r * jb.row_stride];
  }
  red[rl][il] = s;
  __syncthreads();
  if (rl == 0 && c < jb.cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][il];
    int dc = c;
    if (jb.rowmap == 1) dc = geglu_row_unmap(c, jb.F);
    if (dc >= 0 && dc < jb.dst_len) jb.dst[(long)b * jb.dst_bstride + dc] = t;
  }
}
