// RMSNorm / AdaptiveRMSNorm forward+backward and the backward of MultiheadRMSNorm+rotary.
// Memory-bound row kernels: one wave64 per row, float4 (16 B/lane) loads, wave-shuffle reductions.
#include "common.hpp"
#include <stdlib.h>

namespace {

// The row kernels are templated on NC = float4 chunks per lane actually needed (2: D <= 512, 4: D <= 1024, 8: D <= 2048): with
// the arrays sized for D = 2048 the D = 512 instantiation carried 132 (forward) / 218 (backward) VGPRs -- 3 resp. 2 waves per
// SIMD for kernels whose only job is to keep HBM requests in flight.
#define VBX_NC_DISPATCH(D_, CALL) \
  do {                            \
    if ((D_) <= 512) { CALL(2); } else if ((D_) <= 1024) { CALL(4); } else { CALL(8); } \
  } while (0)

// ---------------------------------------------------------------- forward
// y = x / max(|x|, 1e-12) * sqrt(D) * gamma[b] (+ beta[b])        (voicebox_pytorch.py:246-247, 270-276)
// FWD_ROWS = rows per wave in flight (loads of all issued before any reduction)
template <int FWD_ROWS, int NC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, long gb_stride,
                                                           u16* __restrict__ y, u16* __restrict__ y16, int B, int Np, int n0,
                                                           int rpb, int D, float* __restrict__ y32) {
  // one wave per row; the row, its gamma and its beta are all requested before the reduction so that a single memory round
  // trip is exposed per row (FWD_ROWS is kept for the A/B record: two rows per wave in flight measured 9 -> 14 us)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const float sqrtD = sqrtf((float)D);
  const long rows = (long)B * rpb;
  for (long r0 = ((long)blockIdx.x * 4 + wave) * FWD_ROWS; r0 < rows; r0 += (long)gridDim.x * 4 * FWD_ROWS) {
#pragma unroll
    for (int k = 0; k < FWD_ROWS; k++) {
      const long ri = r0 + k;
      if (ri >= rows) break;
      const int b = (int)(ri / rpb), j = (int)(ri - (long)b * rpb);
      const float4* xr = reinterpret_cast<const float4*>(x + ((long)b * Np + n0 + j) * D);
      const float4* g4 = reinterpret_cast<const float4*>(gamma + (long)b * gb_stride);
      const float4* b4 = beta ? reinterpret_cast<const float4*>(beta + (long)b * gb_stride) : nullptr;
      float4 v[NC], g[NC], bt[NC];
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const int c = lane + 64 * i;
        if (c < D4) {
          v[i] = xr[c];
          g[i] = g4[c];
          bt[i] = b4 ? b4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const int c = lane + 64 * i;
        if (c < D4) ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
      }
      ss = wave_sum(ss);
      const float r = sqrtD / fmaxf(sqrtf(ss), 1e-12f);
      uint2* yr = y ? reinterpret_cast<uint2*>(y + ri * D) : nullptr;
      uint2* yr16 = y16 ? reinterpret_cast<uint2*>(y16 + ri * D) : nullptr;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        const int c = lane + 64 * i;
        if (c < D4) {
          const float4 o = make_float4(v[i].x * r * g[i].x + bt[i].x, v[i].y * r * g[i].y + bt[i].y, v[i].z * r * g[i].z + bt[i].z,
                                       v[i].w * r * g[i].w + bt[i].w);
          if (yr) yr[c] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
          if (yr16) yr16[c] = make_uint2(pack_f16x2_sat(o.x, o.y), pack_f16x2_sat(o.z, o.w));
          if (y32) reinterpret_cast<float4*>(y32 + ri * D)[c] = o;
        }
      }
    }
  }
}

// Round 6: the lean row kernel.  The round-1..5 kernel above issues ~600 instructions per row (64-bit per-lane addresses, a 64-bit
// division, six ds_bpermute round trips for the wave sum, compare/select clamps per element, repacking of the conversion results):
// 8320 rows x 600 / 1024 SIMDs x ~4 cycles = 9.7 us -- the launch was INSTRUCTION-bound at 3.2 TB/s, not memory-bound (found by
// giving a wave more rows with the next row prefetched: time grew with rows per wave).  Here: the row is addressed by a scalar base
// (grid = (rows of a batch / 4, B): no division, wave-uniform pointers), D == 256 * NC exactly (no per-chunk guards), the wave sum
// runs on DPP + four v_readlane, conversions take adjacent pairs, and the fp16 saturation is decided once per wave (a ballot on
// max |y| > 65504; NaN never raises it and converts to NaN on the fast path, as before).  Same arithmetic, same results.
template <int CTRL>
VBX_DEV float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
VBX_DEV float wave_sum_dpp(float v) {  // every lane ends with the sum over the 64 lanes
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);  // row_ror:4
  v += dpp_mov<0x128>(v);  // row_ror:8  -> every lane of a 16-lane row holds the row's sum
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}
VBX_DEV unsigned cvt2_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, b2));
}
VBX_DEV unsigned cvt2_f16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, h2));
}
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_lean_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, long gb_stride, u16* __restrict__ y,
                                                                u16* __restrict__ y16, int Np, int n0, int rpb, float* __restrict__ y32) {
  constexpr int D = 256 * NC;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y, j = blockIdx.x * 4 + wave;
  if (j >= rpb) return;
  const float4* xr = reinterpret_cast<const float4*>(x + ((long)b * Np + n0 + j) * D);
  const float4* g4 = reinterpret_cast<const float4*>(gamma + (long)b * gb_stride);
  const float4* b4 = reinterpret_cast<const float4*>(beta + (long)b * gb_stride);
  float4 v[NC], g[NC], bt[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) {
    v[i] = xr[lane + 64 * i];
    g[i] = g4[lane + 64 * i];
    bt[i] = beta ? b4[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
  ss = wave_sum_dpp(ss);
  const float r = sqrtf((float)D) / fmaxf(sqrtf(ss), 1e-12f);
  const long ri = (long)b * rpb + j;
  float4 o[NC];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NC; i++) {
    o[i] = make_float4(v[i].x * r * g[i].x + bt[i].x, v[i].y * r * g[i].y + bt[i].y, v[i].z * r * g[i].z + bt[i].z,
                       v[i].w * r * g[i].w + bt[i].w);
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[i].x), fabsf(o[i].y)), fmaxf(fabsf(o[i].z), fabsf(o[i].w))));
  }
  if (y) {
    uint2* yr = reinterpret_cast<uint2*>(y + ri * D);
#pragma unroll
    for (int i = 0; i < NC; i++) yr[lane + 64 * i] = make_uint2(cvt2_bf16(o[i].x, o[i].y), cvt2_bf16(o[i].z, o[i].w));
  }
  if (y16) {
    uint2* yr16 = reinterpret_cast<uint2*>(y16 + ri * D);
    if (__builtin_amdgcn_ballot_w64(amax > 65504.0f) == 0) {  // nothing to clamp in this row (NaN compares false and converts to NaN)
#pragma unroll
      for (int i = 0; i < NC; i++) yr16[lane + 64 * i] = make_uint2(cvt2_f16(o[i].x, o[i].y), cvt2_f16(o[i].z, o[i].w));
    } else {
#pragma unroll
      for (int i = 0; i < NC; i++) yr16[lane + 64 * i] = make_uint2(pack_f16x2_sat(o[i].x, o[i].y), pack_f16x2_sat(o[i].z, o[i].w));
    }
  }
  if (y32) {
#pragma unroll
    for (int i = 0; i < NC; i++) reinterpret_cast<float4*>(y32 + ri * D)[lane + 64 * i] = o[i];
  }
}

// ---------------------------------------------------------------- backward
// u = x/|x| ; y = sqrt(D) u*gamma + beta
// dgamma[b] += sqrt(D) u*dy ; dbeta[b] += dy ; du = sqrt(D) gamma*dy ; dx = (du - u (u.du)) / |x|
// grid (chunks, B); each block handles RB_ROWS rows of one batch; partials -> part[b][chunk][2][D]
// RB_ROWS rows per block: 16 (default) or 8 (VBX_RMS_BWD_ROWS=8; one row per wave, twice the blocks -- measured in the same
// run: train step 12.75 -> 12.99 ms, the doubled partial records cost more than the extra parallelism buys)
// NB_WAVES = waves per block of the backward kernel (16-row chunk -> 2 rows per wave with 8 waves; 4 waves when the
// [NB_WAVES][3][D] fp32 reduction buffer of 8 waves would exceed the 160 KiB of LDS, i.e. D > 1664)
template <int NB_WAVES, int RB_ROWS, int NC>
__global__ __launch_bounds__(64 * NB_WAVES) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           long gb_stride, const u16* __restrict__ dy,
                                                           const float* __restrict__ dx_in, float* __restrict__ dx_out,
                                                           u16* __restrict__ dxb, float* __restrict__ part,
                                                           float* __restrict__ cpart, int Np, int n0, int rpb, int D) {
  // cpart (optional): per-chunk column sums of dx_in, [b][chunk][D] -- the bias gradient of the Linear whose output was
  // added to the residual stream right after this norm's input (FeedForward[3].bias, voicebox_pytorch.py:348,472)
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NB_WAVES][3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x, b = blockIdx.y, chunks = gridDim.x;
  const int D4 = D >> 2;
  const float sqrtD = sqrtf((float)D);
  const float4* g4 = reinterpret_cast<const float4*>(gamma + (long)b * gb_stride);
  float4 ag[NC], ab[NC], ac[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); ac[i] = make_float4(0, 0, 0, 0); }
  // (Round 5 tried requesting every global load of the wave's rows -- x, dy AND the incoming dx of both rows -- before the first
  //  reduction, to take two dependent memory round trips per row off the wave's chain: 94 VGPRs instead of 60-odd, and the kernel
  //  went from 16.8 to 19.7 us per launch in the train-step profile.  The row-after-row order below stays.)
  for (int k = 0; k < RB_ROWS / NB_WAVES; k++) {
    const int j = chunk * RB_ROWS + wave + NB_WAVES * k;
    if (j >= rpb) break;
    const long xrow = ((long)b * Np + n0 + j) * D;
    const long drow = ((long)b * rpb + j) * D;
    const float4* xr = reinterpret_cast<const float4*>(x + xrow);
    const uint2* dyr = reinterpret_cast<const uint2*>(dy + drow);
    float4 xv[NC], dv[NC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        xv[i] = xr[c];
        const uint2 p = dyr[c];
        dv[i] = make_float4(bf16_to_f32((u16)(p.x & 0xffff)), bf16_to_f32((u16)(p.x >> 16)),
                            bf16_to_f32((u16)(p.y & 0xffff)), bf16_to_f32((u16)(p.y >> 16)));
        ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
      }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        const float4 g = g4[c];
        // u
        xv[i].x *= inv; xv[i].y *= inv; xv[i].z *= inv; xv[i].w *= inv;
        ag[i].x += sqrtD * xv[i].x * dv[i].x; ag[i].y += sqrtD * xv[i].y * dv[i].y;
        ag[i].z += sqrtD * xv[i].z * dv[i].z; ag[i].w += sqrtD * xv[i].w * dv[i].w;
        ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
        // du
        dv[i].x *= sqrtD * g.x; dv[i].y *= sqrtD * g.y; dv[i].z *= sqrtD * g.z; dv[i].w *= sqrtD * g.w;
        dot += xv[i].x * dv[i].x + xv[i].y * dv[i].y + xv[i].z * dv[i].z + xv[i].w * dv[i].w;
      }
    }
    dot = wave_sum(dot);
    const float4* din = dx_in ? reinterpret_cast<const float4*>(dx_in + xrow) : nullptr;
    float4* dout = reinterpret_cast<float4*>(dx_out + xrow);
    uint2* dbo = dxb ? reinterpret_cast<uint2*>(dxb + xrow) : nullptr;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        float4 o = make_float4((dv[i].x - xv[i].x * dot) * inv, (dv[i].y - xv[i].y * dot) * inv,
                               (dv[i].z - xv[i].z * dot) * inv, (dv[i].w - xv[i].w * dot) * inv);
        if (din) {
          const float4 a = din[c];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
          ac[i].x += a.x; ac[i].y += a.y; ac[i].z += a.z; ac[i].w += a.w;
        }
        dout[c] = o;
        if (dbo) dbo[c] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
      }
    }
  }
  // cross-wave reduction of the gamma/beta partials
  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int c = lane + 64 * i;
    if (c < D4) {
      r4[(wave * 3 + 0) * D4 + c] = ag[i];
      r4[(wave * 3 + 1) * D4 + c] = ab[i];
      r4[(wave * 3 + 2) * D4 + c] = ac[i];
    }
  }
  __syncthreads();
  float4* p4 = reinterpret_cast<float4*>(part + ((long)b * chunks + chunk) * 2 * D);
  float4* c4 = cpart ? reinterpret_cast<float4*>(cpart + ((long)b * chunks + chunk) * D) : nullptr;
  for (int idx = threadIdx.x; idx < 3 * D4; idx += 64 * NB_WAVES) {
    const int which = idx / D4, c = idx - which * D4;
    if (which == 2 && !c4) continue;
    float4 s = r4[(0 * 3 + which) * D4 + c];
#pragma unroll
    for (int w = 1; w < NB_WAVES; w++) {
      const float4 t = r4[(w * 3 + which) * D4 + c];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (which < 2) p4[which * D4 + c] = s;
    else c4[c] = s;
  }
}

// Round 6: the lean form of the backward row kernel for D == 256 * NC: scalar row bases (no per-lane 64-bit addresses), no per-chunk
// guards, gamma loaded once per workgroup, the incoming dx requested together with x and dy (one memory round trip per row instead
// of three dependent ones), DPP wave sums.  Same arithmetic and partial-record layout as rmsnorm_bwd_kernel<8, 16, NC>.
template <int NC>
__global__ __launch_bounds__(512) void rmsnorm_bwd_lean_kernel(const float* __restrict__ x, const float* __restrict__ gamma, long gb_stride,
                                                                const u16* __restrict__ dy, const float* __restrict__ dx_in,
                                                                float* __restrict__ dx_out, u16* __restrict__ dxb, float* __restrict__ part,
                                                                float* __restrict__ cpart, int Np, int n0, int rpb) {
  constexpr int D = 256 * NC, D4 = D / 4, NB_WAVES = 8, RB_ROWS = 16;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [8][3][D]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chunk = blockIdx.x, b = blockIdx.y, chunks = gridDim.x;
  const float sqrtD = sqrtf((float)D);
  const float4* g4 = reinterpret_cast<const float4*>(gamma + (long)b * gb_stride);
  float4 g[NC], ag[NC], ab[NC], ac[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) {
    g[i] = g4[lane + 64 * i];
    ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); ac[i] = make_float4(0, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < RB_ROWS / NB_WAVES; k++) {
    const int j = chunk * RB_ROWS + wave + NB_WAVES * k;
    if (j >= rpb) break;
    const long xrow = ((long)b * Np + n0 + j) * D;
    const long drow = ((long)b * rpb + j) * D;
    const float4* xr = reinterpret_cast<const float4*>(x + xrow);
    const uint2* dyr = reinterpret_cast<const uint2*>(dy + drow);
    const float4* din = reinterpret_cast<const float4*>(dx_in + xrow);
    float4 xv[NC], dv[NC], a[NC];
    uint2 praw[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) {
      xv[i] = xr[lane + 64 * i];
      praw[i] = dyr[lane + 64 * i];
      a[i] = dx_in ? din[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) ss += xv[i].x * xv[i].x + xv[i].y * xv[i].y + xv[i].z * xv[i].z + xv[i].w * xv[i].w;
    ss = wave_sum_dpp(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      dv[i] = make_float4(bf16_to_f32((u16)(praw[i].x & 0xffff)), bf16_to_f32((u16)(praw[i].x >> 16)),
                          bf16_to_f32((u16)(praw[i].y & 0xffff)), bf16_to_f32((u16)(praw[i].y >> 16)));
      xv[i].x *= inv; xv[i].y *= inv; xv[i].z *= inv; xv[i].w *= inv;
      ag[i].x += sqrtD * xv[i].x * dv[i].x; ag[i].y += sqrtD * xv[i].y * dv[i].y;
      ag[i].z += sqrtD * xv[i].z * dv[i].z; ag[i].w += sqrtD * xv[i].w * dv[i].w;
      ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
      dv[i].x *= sqrtD * g[i].x; dv[i].y *= sqrtD * g[i].y; dv[i].z *= sqrtD * g[i].z; dv[i].w *= sqrtD * g[i].w;
      dot += xv[i].x * dv[i].x + xv[i].y * dv[i].y + xv[i].z * dv[i].z + xv[i].w * dv[i].w;
    }
    dot = wave_sum_dpp(dot);
    float4* dout = reinterpret_cast<float4*>(dx_out + xrow);
    uint2* dbo = reinterpret_cast<uint2*>(dxb + xrow);
#pragma unroll
    for (int i = 0; i < NC; i++) {
      float4 o = make_float4((dv[i].x - xv[i].x * dot) * inv, (dv[i].y - xv[i].y * dot) * inv,
                             (dv[i].z - xv[i].z * dot) * inv, (dv[i].w - xv[i].w * dot) * inv);
      if (dx_in) {
        o.x += a[i].x; o.y += a[i].y; o.z += a[i].z; o.w += a[i].w;
        ac[i].x += a[i].x; ac[i].y += a[i].y; ac[i].z += a[i].z; ac[i].w += a[i].w;
      }
      dout[lane + 64 * i] = o;
      if (dxb) dbo[lane + 64 * i] = make_uint2(cvt2_bf16(o.x, o.y), cvt2_bf16(o.z, o.w));
    }
  }
  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int c = lane + 64 * i;
    r4[(wave * 3 + 0) * D4 + c] = ag[i];
    r4[(wave * 3 + 1) * D4 + c] = ab[i];
    r4[(wave * 3 + 2) * D4 + c] = ac[i];
  }
  __syncthreads();
  float4* p4 = reinterpret_cast<float4*>(part + ((long)b * chunks + chunk) * 2 * D);
  float4* c4 = cpart ? reinterpret_cast<float4*>(cpart + ((long)b * chunks + chunk) * D) : nullptr;
  for (int idx = threadIdx.x; idx < 3 * D4; idx += 64 * NB_WAVES) {
    const int which = idx / D4, c = idx - which * D4;
    if (which == 2 && !c4) continue;
    float4 s = r4[(0 * 3 + which) * D4 + c];
#pragma unroll
    for (int w = 1; w < NB_WAVES; w++) {
      const float4 t = r4[(w * 3 + which) * D4 + c];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (which < 2) p4[which * D4 + c] = s;
    else c4[c] = s;
  }
}

// out[b][which][d] = sum_chunk part[b][chunk][which][d]   (optionally also summed over b)
// block = 64 idx x 4 chunk lanes; grid (2D/64, sum_batch ? 1 : B)
__global__ __launch_bounds__(256) void reduce_norm_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                    long out_b_stride, int B, int chunks, int W, int sum_batch) {
  // W = row width of the partial records (2*D for gamma|beta records, D for column-sum records)
  __shared__ float red[4][64];
  const int il = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + il;
  float s = 0.f;
  if (idx < W) {
    if (sum_batch) {
      const long total = (long)B * chunks;
      long c = cl;
      for (; c + 12 < total; c += 16) {  // four records in flight (same summation order)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = part[(c + 4 * u) * W + idx];
#pragma unroll
        for (int u = 0; u < 4; u++) s += v[u];
      }
      for (; c < total; c += 4) s += part[c * W + idx];
    } else {
      const int b = blockIdx.y;
      int c = cl;
      for (; c + 12 < chunks; c += 16) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = part[((long)b * chunks + c + 4 * u) * W + idx];
#pragma unroll
        for (int u = 0; u < 4; u++) s += v[u];
      }
      for (; c < chunks; c += 4) s += part[((long)b * chunks + c) * W + idx];
    }
  }
  red[cl][il] = s;
  __syncthreads();
  if (cl == 0 && idx < W) {
    const float t = red[0][il] + red[1][il] + red[2][il] + red[3][il];
    if (sum_batch) out[idx] = t;
    else out[(long)blockIdx.y * out_b_stride + idx] = t;
  }
}

// ---------------------------------------------------------------- MultiheadRMSNorm + rotary backward
// forward (per (b,h,n), vector t in R^64):  u = t/|t| ; y = u * qk_scale * gamma[h] ; qhat = R(n) y
// R(n) y [d] = y[d] c[d%32] + (d<32 ? -y[d+32] : y[d-32]) s[d%32]
// grid (H, B, 2*NSPLIT): z = which*NSPLIT + split.  8 lanes per row, 8 d's per lane.
constexpr int NSPLIT = 4;
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ dk,
                                                              const u16* __restrict__ q16, const u16* __restrict__ k16,
                                                              const float* __restrict__ qrn, const float* __restrict__ krn,
                                                              const float* __restrict__ qg, const float* __restrict__ kg,
                                                              const float* __restrict__ rc, const float* __restrict__ rs,
                                                              float qk_scale, u16* __restrict__ dqkv, int ld,
                                                              float* __restrict__ gpart, int B, int H, int Np, float q16_inv) {
  __shared__ float red[32][64];
  const int h = blockIdx.x, b = blockIdx.y;
  const int which = blockIdx.z / NSPLIT, split = blockIdx.z % NSPLIT;
  const float* dsrc = which == 0 ? dq : dk;
  const u16* hsrc = which == 0 ? q16 : k16;
  const float* rn = which == 0 ? qrn : krn;
  const float* gam = (which == 0 ? qg : kg);
  const float xh_inv = which == 0 ? q16_inv : 1.0f;  // q16 carries scale * log2(e) (include/vbx.h, attention section)
  const int sub = threadIdx.x & 7, slot = threadIdx.x >> 3;  // 32 row slots
  const int d0 = sub * 8;
  const bool lowhalf = d0 < 32;
  const long bh = (long)b * H + h;
  const int per = (Np + NSPLIT - 1) / NSPLIT;
  const int nbeg = split * per, nend = min(Np, nbeg + per);
  float gacc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) gacc[i] = 0.f;
  float gm[8];
#pragma unroll
  for (int i = 0; i < 8; i++) gm[i] = (qk_scale > 0.f) ? gam[h * 64 + d0 + i] : 1.f;

  for (int nb = nbeg; nb < nend; nb += 32) {
    const int n = nb + slot;
    const bool valid = n < nend;
    const int nc = valid ? n : (nend - 1);
    const long ro = (bh * Np + nc) * 64 + d0;
    float g[8], qh[8];
    {
      const float4 a = *reinterpret_cast<const float4*>(dsrc + ro);
      const float4 c = *reinterpret_cast<const float4*>(dsrc + ro + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = c.x; g[5] = c.y; g[6] = c.z; g[7] = c.w;
      const uint4 hq = *reinterpret_cast<const uint4*>(hsrc + ro);
      const unsigned w[4] = {hq.x, hq.y, hq.z, hq.w};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        qh[2 * i] = f16_to_f32((u16)(w[i] & 0xffff));
        qh[2 * i + 1] = f16_to_f32((u16)(w[i] >> 16));
      }
    }
    const float* cp = rc + (long)nc * 32 + (d0 & 31);
    const float* sp = rs + (long)nc * 32 + (d0 & 31);
    float dy[8], yv[8];
    const float sgn = lowhalf ? 1.f : -1.f;  // transpose rotation
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float gp = __shfl_xor(g[i], 4, 64);
      const float qp = __shfl_xor(qh[i], 4, 64);
      dy[i] = g[i] * cp[i] + sgn * gp * sp[i];
      yv[i] = (qh[i] * cp[i] + sgn * qp * sp[i]) * xh_inv;
    }
    float out[8];
    if (qk_scale > 0.f) {
      const float rinv = rn[bh * Np + nc];
      float u[8], du[8], dot = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float sg = qk_scale * gm[i];
        u[i] = (fabsf(sg) > 1e-20f) ? yv[i] / sg : 0.f;
        if (valid) gacc[i] += dy[i] * u[i] * qk_scale;
        du[i] = dy[i] * sg;
        dot += u[i] * du[i];
      }
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot += __shfl_xor(dot, 4, 64);
#pragma unroll
      for (int i = 0; i < 8; i++) out[i] = (du[i] - u[i] * dot) * rinv;
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) out[i] = dy[i];
    }
    if (valid) {
      u16* o = dqkv + ((long)b * Np + n) * ld + which * H * 64 + h * 64 + d0;
      *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf16x2(out[0], out[1]), pack_bf16x2(out[2], out[3]),
                                                pack_bf16x2(out[4], out[5]), pack_bf16x2(out[6], out[7]));
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) red[slot][d0 + i] = gacc[i];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int r = 0; r < 32; r++) s += red[r][threadIdx.x];
    gpart[(((long)which * B * NSPLIT + (long)b * NSPLIT + split) * H + h) * 64 + threadIdx.x] = s;
  }
}

}  // namespace

static int rmsnorm_fwd_launch(const float* x, const float* gamma, const float* beta, long gb_stride, void* y_bf16, void* y_f16,
                              float* y_f32, int B, int Np, int n0, int rows_per_batch, int D, void* stream) {
  VBX_REQUIRE(x && gamma && (y_bf16 || y_f16 || y_f32), "vbx_rmsnorm_fwd: null pointer");
  VBX_REQUIRE(D % 4 == 0 && D <= 2048 && D > 0, "vbx_rmsnorm_fwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
  VBX_REQUIRE(B > 0 && rows_per_batch > 0 && n0 >= 0 && n0 + rows_per_batch <= Np, "vbx_rmsnorm_fwd: bad row range");
  const long rows = (long)B * rows_per_batch;
  static const int rpw = getenv("VBX_RMS_ROWS") ? atoi(getenv("VBX_RMS_ROWS")) : 1;  // A/B: rows per wave in flight (2: 9 -> 14 us)
  int blocks = cdiv(rows, 4 * (rpw == 2 ? 2 : 1));
  if (blocks > 4096) blocks = 4096;
#define VBX_RF_LAUNCH2(NC_)                                                                                                      \
  hipLaunchKernelGGL((rmsnorm_fwd_kernel<2, NC_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, gb_stride, \
                     (u16*)y_bf16, (u16*)y_f16, B, Np, n0, rows_per_batch, D, y_f32)
#define VBX_RF_LAUNCH1(NC_)                                                                                                      \
  hipLaunchKernelGGL((rmsnorm_fwd_kernel<1, NC_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, gb_stride, \
                     (u16*)y_bf16, (u16*)y_f16, B, Np, n0, rows_per_batch, D, y_f32)
  static const bool lean = !(getenv("VBX_RMS_LEAN") && atoi(getenv("VBX_RMS_LEAN")) == 0);  // 0: A/B against the generic kernel
  if (lean && (D == 512 || D == 1024 || D == 2048)) {
    const dim3 lgrid(cdiv(rows_per_batch, 4), B);
#define VBX_RF_LAUNCHL(NC_)                                                                                                          \
  hipLaunchKernelGGL((rmsnorm_fwd_lean_kernel<NC_>), lgrid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, gb_stride, (u16*)y_bf16, \
                     (u16*)y_f16, Np, n0, rows_per_batch, y_f32)
    if (D == 512) VBX_RF_LAUNCHL(2);
    else if (D == 1024) VBX_RF_LAUNCHL(4);
    else VBX_RF_LAUNCHL(8);
#undef VBX_RF_LAUNCHL
    VBX_LAUNCH_CHECK();
    return 0;
  }
  if (rpw == 2) VBX_NC_DISPATCH(D, VBX_RF_LAUNCH2);
  else VBX_NC_DISPATCH(D, VBX_RF_LAUNCH1);
#undef VBX_RF_LAUNCH1
#undef VBX_RF_LAUNCH2
  VBX_LAUNCH_CHECK();
  return 0;
}
// every output at once (precise.hip: fp32 rows for the hi/lo split + the bf16 copy the backward reads)
int vbx_rmsnorm_fwd_multi(const float* x, const float* gamma, const float* beta, long gb_stride, void* y_bf16, void* y_f16,
                          float* y_f32, int B, int Np, int n0, int rows_per_batch, int D, void* stream) {
  return rmsnorm_fwd_launch(x, gamma, beta, gb_stride, y_bf16, y_f16, y_f32, B, Np, n0, rows_per_batch, D, stream);
}
extern "C" int vbx_rmsnorm_fwd(const float* x, const float* gamma, const float* beta, long gb_stride, void* y_bf16,
                               void* y_f16, int B, int Np, int n0, int rows_per_batch, int D, void* stream) {
  return rmsnorm_fwd_launch(x, gamma, beta, gb_stride, y_bf16, y_f16, nullptr, B, Np, n0, rows_per_batch, D, stream);
}
extern "C" int vbx_rmsnorm_fwd_f32(const float* x, const float* gamma, const float* beta, long gb_stride, float* y_f32, int B,
                                   int Np, int n0, int rows_per_batch, int D, void* stream) {
  return rmsnorm_fwd_launch(x, gamma, beta, gb_stride, nullptr, nullptr, y_f32, B, Np, n0, rows_per_batch, D, stream);
}

static int rb_rows() {
  static const int r = [] {
    const int v = getenv("VBX_RMS_BWD_ROWS") ? atoi(getenv("VBX_RMS_BWD_ROWS")) : 16;
    return (v == 8 || v == 32) ? v : 16;
  }();
  return r;
}
extern "C" int vbx_rmsnorm_bwd_chunks(int rows_per_batch) { return cdiv(rows_per_batch, rb_rows()); }

extern "C" int vbx_rmsnorm_bwd(const float* x, const float* gamma, long gb_stride, const void* dy_bf16, const float* dx_in,
                               float* dx_out, void* dxb_bf16, float* part, float* colpart, int B, int Np, int n0,
                               int rows_per_batch, int D, void* stream) {
  VBX_REQUIRE(x && gamma && dy_bf16 && dx_out && part, "vbx_rmsnorm_bwd: null pointer");
  VBX_REQUIRE(D % 4 == 0 && D <= 2048 && D > 0, "vbx_rmsnorm_bwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
  VBX_REQUIRE(B > 0 && rows_per_batch > 0 && n0 >= 0 && n0 + rows_per_batch <= Np, "vbx_rmsnorm_bwd: bad row range");
  dim3 grid(cdiv(rows_per_batch, rb_rows()), B);
  VBX_REQUIRE(!colpart || dx_in, "vbx_rmsnorm_bwd: column sums need dx_in");
  const bool eight = (size_t)8 * 3 * D * sizeof(float) <= 160 * 1024;
  const size_t lds = (size_t)(eight ? 8 : 4) * 3 * D * sizeof(float);
  static const bool lean = !(getenv("VBX_RMS_LEAN") && atoi(getenv("VBX_RMS_LEAN")) == 0);  // 0: A/B against the generic kernel
  if (lean && rb_rows() == 16 && (D == 512 || D == 1024)) {  // (D = 2048: the [8][3][D] reduction buffer exceeds the LDS, generic path)
#define VBX_RB_LAUNCHL(NC_)                                                                                                          \
  do {                                                                                                                               \
    static bool attr_ = false;                                                                                                       \
    if (!attr_) {                                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rmsnorm_bwd_lean_kernel<NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                                                         \
      attr_ = true;                                                                                                                  \
    }                                                                                                                                \
    hipLaunchKernelGGL((rmsnorm_bwd_lean_kernel<NC_>), grid, dim3(512), lds, (hipStream_t)stream, x, gamma, gb_stride,                 \
                       (const u16*)dy_bf16, dx_in, dx_out, (u16*)dxb_bf16, part, colpart, Np, n0, rows_per_batch);                   \
  } while (0)
    if (D == 512) VBX_RB_LAUNCHL(2);
    else VBX_RB_LAUNCHL(4);
#undef VBX_RB_LAUNCHL
    VBX_LAUNCH_CHECK();
    return 0;
  }
#define VBX_RB_LAUNCH_NC(W, R, NC_)                                                                                             \
  do {                                                                                                                         \
    static bool attr_ = false;                                                                                                 \
    if (!attr_) {                                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rmsnorm_bwd_kernel<W, R, NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                                                   \
      attr_ = true;                                                                                                            \
    }                                                                                                                          \
    hipLaunchKernelGGL((rmsnorm_bwd_kernel<W, R, NC_>), grid, dim3(64 * W), lds, (hipStream_t)stream, x, gamma, gb_stride,       \
                       (const u16*)dy_bf16, dx_in, dx_out, (u16*)dxb_bf16, part, colpart, Np, n0, rows_per_batch, D);          \
  } while (0)
#define VBX_RB_LAUNCH(W, R)                                                 \
  do {                                                                     \
    if (D <= 512) VBX_RB_LAUNCH_NC(W, R, 2);                               \
    else if (D <= 1024) VBX_RB_LAUNCH_NC(W, R, 4);                         \
    else VBX_RB_LAUNCH_NC(W, R, 8);                                        \
  } while (0)
  if (eight && rb_rows() == 8) VBX_RB_LAUNCH(8, 8);
  else if (eight && rb_rows() == 32) VBX_RB_LAUNCH(8, 32);
  else if (eight) VBX_RB_LAUNCH(8, 16);
  else if (rb_rows() == 8) VBX_RB_LAUNCH(4, 8);
  else if (rb_rows() == 32) VBX_RB_LAUNCH(4, 32);
  else VBX_RB_LAUNCH(4, 16);
#undef VBX_RB_LAUNCH_NC
#undef VBX_RB_LAUNCH
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_reduce_norm_partials(const float* part, float* out, long out_b_stride, int B, int chunks, int D,
                                        int sum_batch, void* stream) {
  VBX_REQUIRE(part && out && B > 0 && chunks > 0 && D > 0, "vbx_reduce_norm_partials: bad args");
  dim3 grid(cdiv(2 * D, 64), sum_batch ? 1 : B);
  hipLaunchKernelGGL(reduce_norm_partials_kernel, grid, dim3(256), 0, (hipStream_t)stream, part, out, out_b_stride, B,
                     chunks, 2 * D, sum_batch);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_reduce_col_partials(const float* colpart, float* out, float* tmp /* [B][D] */, int B, int chunks, int D,
                                       void* stream) {
  VBX_REQUIRE(colpart && out && tmp && B > 0 && chunks > 0 && D > 0, "vbx_reduce_col_partials: bad args");
  // per-batch sums over the chunks (B x D/64 blocks), then the few batch rows
  hipLaunchKernelGGL(reduce_norm_partials_kernel, dim3(cdiv(D, 64), B), dim3(256), 0, (hipStream_t)stream, colpart, tmp, (long)D,
                     B, chunks, D, 0);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(reduce_norm_partials_kernel, dim3(cdiv(D, 64), 1), dim3(256), 0, (hipStream_t)stream, tmp, out, 0L, B, 1, D,
                     1);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_qknorm_rope_bwd_gpart_rows(int B) { return B * NSPLIT; }

extern "C" int vbx_qknorm_rope_bwd(const float* dq, const float* dk, const void* q16, const void* k16, const float* q_rnorm,
                                   const float* k_rnorm, const float* q_gamma, const float* k_gamma, const float* rot_cos,
                                   const float* rot_sin, float qk_scale, void* dqkv, int ld, float* gpart, int B, int H,
                                   int Np, float q16_scale, void* stream) {
  VBX_REQUIRE(dq && dk && q16 && k16 && rot_cos && rot_sin && dqkv && gpart, "vbx_qknorm_rope_bwd: null pointer");
  VBX_REQUIRE(qk_scale <= 0.f || (q_rnorm && k_rnorm && q_gamma && k_gamma), "vbx_qknorm_rope_bwd: qk-norm needs stats");
  VBX_REQUIRE(ld % 8 == 0 && q16_scale > 0.f, "vbx_qknorm_rope_bwd: ld must be a multiple of 8, q16_scale > 0");
  dim3 grid(H, B, 2 * NSPLIT);
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dq, dk, (const u16*)q16,
                     (const u16*)k16, q_rnorm, k_rnorm, q_gamma, k_gamma, rot_cos, rot_sin, qk_scale, (u16*)dqkv, ld, gpart,
                     B, H, Np, 1.0f / q16_scale);
  VBX_LAUNCH_CHECK();
  return 0;
}
