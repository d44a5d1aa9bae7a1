// Exact-operand ("precise") forward: the same VoiceBox forward with every matrix product evaluated to fp32 accuracy.
//
// Why it exists.  The fast path rounds every forward GEMM / attention operand to fp16 (2^-11).  At the reference's own
// initialisation the qk-normed logits 10 q.k have std ~80, the softmax is nearly one-hot and the 12-layer map is chaotic: each
// rounded operand class moves the loss by O(1e-3), so the fast path holds the north star's 1e-3 only where the
// problem is well conditioned.  This mode removes the operand rounding without a second GEMM implementation:
//
//  * every forward GEMM runs through the SAME vbx_gemm tiles (fp16 MFMA, fp32 accumulate) with both operands split into
//    fp16 hi + lo parts and concatenated along K:   A' = [A_hi | A_hi | A_lo],  W' = [W_hi | W_lo | W_hi],  K' = 3K, so that
//    A'.W'^T = A_hi W_hi + A_hi W_lo + A_lo W_hi  (the dropped A_lo W_lo term is 2^-22 relative; the lo segments carry a
//    factor 2^8 and their partners 2^-8 so that no lo part is an fp16 subnormal, see split_hi_lo).  fp16 x fp16 products are exact in
//    the MFMA's fp32 accumulator, hence the result carries ~22 operand bits -- fp32-class -- at 3x the MFMA work;
//  * what the fused QKV / GEGLU epilogues do on fp16 outputs is done here by small fp32 kernels on the fp32 GEMM result
//    (qk-norm + rotary, erf-GELU gate), which also write the fp16 / bf16 copies the (unchanged, bf16-operand) backward reads;
//  * attention is a plain fp32 FMA flash kernel (q, k, v, P all fp32; online softmax in exp2) -- no MFMA, speed is irrelevant here;
//  * the adaLN projections read the fp32 master weights.
//
// Reference call sites: voicebox_pytorch.py:987-1115 (VoiceBox.forward), :412-479 (Transformer.forward), attend.py:121-135.
// Selected per model by vbx_model.precise (engine.py: VBX_PRECISE=1 / voicebox_pytorch_amd.precise_mode()).
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <vector>

int vbx_rmsnorm_fwd_multi(const float* x, const float* gamma, const float* beta, long gb_stride, void* y_bf16, void* y_f16,
                          float* y_f32, int B, int Np, int n0, int rows_per_batch, int D, void* stream);  // norm.hip

namespace {

#define ST ((hipStream_t)stream)
inline int grid_for(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

// v = hi + lo with hi = fp16(v) and lo = v - hi (exact in fp32), stored as fp16(lo * 2^8): unscaled, the lo part of an nn.Linear
// weight (|w| <= K^-0.5: lo ~ 2^-11 w ~ 1e-5) is an fp16 SUBNORMAL and the pair carries 19 bits instead of 22.  The product terms
// are unchanged because the partner segment carries 2^-8: A' = [A_hi | A_hi 2^-8 | A_lo 2^8], W' = [W_hi | W_lo 2^8 | W_hi 2^-8].
// (The 2^-8 copies of the hi parts only meet lo parts, so what they lose to underflow is <= 2^-16 * |lo| absolute: nothing.)
constexpr float P3_UP = 256.0f, P3_DOWN = 1.0f / 256.0f;
VBX_DEV void split_hi_lo(float v, u16& hi, u16& hi_dn, u16& lo_up) {
  hi = f32_to_f16_sat(v);
  const float h = f16_to_f32(hi);
  hi_dn = f32_to_f16(h * P3_DOWN);
  lo_up = f32_to_f16_sat((v - h) * P3_UP);
}

// dst [rows, 3*Kp] = [hi | hi 2^-8 | lo 2^8] of src [rows, K] (row stride ld), columns K..Kp zero
__global__ void split3_kernel(const float* __restrict__ src, long rows, int K, long ld, u16* __restrict__ dst, int Kp) {
  const int cpr = Kp / 4;
  const long total = rows * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c + 3 < K) {
      const float4 t = *reinterpret_cast<const float4*>(src + r * ld + c);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
      for (int e = 0; e < 4; e++)
        if (c + e < K) v[e] = src[r * ld + c + e];
    }
    u16 h[4], hd[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; e++) split_hi_lo(v[e], h[e], hd[e], l[e]);
    const uint2 hv = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    const uint2 dv = make_uint2((unsigned)hd[0] | ((unsigned)hd[1] << 16), (unsigned)hd[2] | ((unsigned)hd[3] << 16));
    const uint2 lv = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    u16* d = dst + r * 3 * (long)Kp + c;
    *reinterpret_cast<uint2*>(d) = hv;
    *reinterpret_cast<uint2*>(d + Kp) = dv;
    *reinterpret_cast<uint2*>(d + 2 * Kp) = lv;
  }
}

// dst [dst_rows, 3*dst_cols] = [hi | lo 2^8 | hi 2^-8] of the (row-mapped, zero-padded) weight, same row map as pack_weight_kernel
__global__ void pack_weight3_kernel(const float* __restrict__ src, int src_rows, int src_cols, u16* __restrict__ dst, int dst_rows,
                                    int dst_cols, int rowmap, int F) {
  const long total = (long)dst_rows * dst_cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i / dst_cols), c = (int)(i - (long)p * dst_cols);
    int r = p;
    if (rowmap == 1) r = geglu_row_unmap(p, F);
    float v = 0.f;
    if (r >= 0 && r < src_rows && c < src_cols) v = src[(long)r * src_cols + c];
    u16 hi, hi_dn, lo_up;
    split_hi_lo(v, hi, hi_dn, lo_up);
    u16* d = dst + (long)p * 3 * dst_cols + c;
    d[0] = hi;
    d[dst_cols] = lo_up;
    d[2 * dst_cols] = hi_dn;
  }
}

// fp32 cat(x, cond * ~cond_mask) [rows, 2D]   (voicebox_pytorch.py:1035,1075-1076)
__global__ void embed_cat_kernel(const float* __restrict__ x, const float* __restrict__ cond, const uint8_t* __restrict__ cmask,
                                 float* __restrict__ out, long rows, int D) {
  const int cpr = D / 4;
  const long total = rows * 2 * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (2 * cpr);
    const int c = (int)(i - row * 2 * cpr);
    const bool second = c >= cpr;
    const int d = (second ? c - cpr : c) * 4;
    float4 v = *reinterpret_cast<const float4*>((second ? cond : x) + row * D + d);
    if (second && cmask && cmask[row]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(out + row * 2 * D + (second ? D : 0) + d) = v;
  }
}

// MultiheadRMSNorm + rotary on the fp32 to_qkv output (voicebox_pytorch.py:320-328, 286-287, 193-199): one wave per (row, head),
// lane = channel.  Same operation order as gemm.hip::EpiQKV.  raw [M, 3*H*64]; outputs head-major [B,H,Np,64].
__global__ __launch_bounds__(256) void qknorm_rope_f32_kernel(const float* __restrict__ raw, int Np, int H, long M, float qk_scale,
                                                              const float* __restrict__ qg, const float* __restrict__ kg,
                                                              const float* __restrict__ rc, const float* __restrict__ rs,
                                                              float* __restrict__ q32, float* __restrict__ k32, float* __restrict__ v32,
                                                              u16* __restrict__ q16, u16* __restrict__ k16, u16* __restrict__ qb,
                                                              u16* __restrict__ kb, u16* __restrict__ vb, u16* __restrict__ v16,
                                                              float* __restrict__ qrn, float* __restrict__ krn, float q16_scale) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= M * H) return;
  const long r = item / H;
  const int h = (int)(item - r * H);
  const int b = (int)(r / Np), n = (int)(r - (long)b * Np);
  const int I = H * 64;
  const long o = (((long)b * H + h) * Np + n) * 64 + lane;
  const long st = ((long)b * H + h) * Np + n;
  const int dc = lane & 31;
  const float c = rc[(long)n * 32 + dc], s = rs[(long)n * 32 + dc];
#pragma unroll
  for (int which = 0; which < 2; which++) {
    float t = raw[r * 3 * I + (long)which * I + h * 64 + lane];
    const float ss = wave_sum(t * t);
    const float rinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (qk_scale > 0.f) t = t * (rinv * qk_scale) * (which == 0 ? qg : kg)[h * 64 + lane];
    const float p = __shfl_xor(t, 32, 64);
    const float out = lane < 32 ? t * c - p * s : t * c + p * s;
    (which == 0 ? q32 : k32)[o] = out;
    u16* d16 = which == 0 ? q16 : k16;
    if (d16) d16[o] = f32_to_f16(which == 0 ? out * q16_scale : out);  // q16 carries scale * log2 e (include/vbx.h, attention)
    u16* db = which == 0 ? qb : kb;
    if (db) db[o] = f32_to_bf16(out);
    float* rn = which == 0 ? qrn : krn;
    if (rn && lane == 0) rn[st] = rinv;
  }
  const float v = raw[r * 3 * I + 2L * I + h * 64 + lane];
  v32[o] = v;
  if (vb) vb[o] = f32_to_bf16(v);
  if (v16) v16[o] = f32_to_f16_sat(v);
}

// Attend.forward (attend.py:121-135) in fp32: softmax(scale q k^T + key mask) v, flash-style with a lane per query row.
// Block = 256 threads = 256 queries of one (b, h); keys stream through LDS in tiles of 64, scores in register chunks of 16.
constexpr int PA_KT = 64, PA_CH = 16;
template <bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const uint8_t* __restrict__ mask,
                                                           float* __restrict__ o32, u16* __restrict__ o16, u16* __restrict__ ob,
                                                           float* __restrict__ lse, int H, int Np, float scale_log2e,
                                                           const unsigned* __restrict__ bits_rm, int W2, float rkeep) {
  __shared__ __attribute__((aligned(16))) float Ks[PA_KT][64];
  __shared__ __attribute__((aligned(16))) float Vs[PA_KT][64];
  __shared__ float valid_s[PA_KT];
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool qok = qi < Np;
  const float* qp = q + ((long)bh * Np + (qok ? qi : 0)) * 64;
  float qr[64], acc[64];
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qp + d);
    qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
    acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
  }
  float m = -1e30f, l = 0.f;
  // attention dropout (attend.py:131): keep bit (key % 32) of word key / 32 of this query's row (ops.hip::attn_dropout_bits_kernel)
  const unsigned* brow = DROP ? bits_rm + ((long)bh * Np + (qok ? qi : 0)) * W2 : nullptr;
  for (int k0 = 0; k0 < Np; k0 += PA_KT) {
    __syncthreads();
    for (int i = threadIdx.x; i < PA_KT * 16; i += 256) {
      const int j = i >> 4, c = (i & 15) * 4;
      const bool ok = k0 + j < Np;
      const long src = ((long)bh * Np + (ok ? k0 + j : 0)) * 64 + c;
      *reinterpret_cast<float4*>(&Ks[j][c]) = *reinterpret_cast<const float4*>(k + src);
      *reinterpret_cast<float4*>(&Vs[j][c]) = *reinterpret_cast<const float4*>(v + src);
    }
    if (threadIdx.x < PA_KT) {
      const int kj = k0 + threadIdx.x;
      valid_s[threadIdx.x] = (kj < Np && (!mask || mask[(long)b * Np + kj])) ? 1.f : 0.f;
    }
    __syncthreads();
    for (int c0 = 0; c0 < PA_KT; c0 += PA_CH) {
      float s[PA_CH];
      float cmax = -1e30f;
#pragma unroll
      for (int j = 0; j < PA_CH; j++) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
          const float4 kk = *reinterpret_cast<const float4*>(&Ks[c0 + j][d]);
          a0 = fmaf(qr[d], kk.x, a0); a1 = fmaf(qr[d + 1], kk.y, a1);
          a0 = fmaf(qr[d + 2], kk.z, a0); a1 = fmaf(qr[d + 3], kk.w, a1);
        }
        const float sv = valid_s[c0 + j] != 0.f ? (a0 + a1) * scale_log2e : -1e30f;
        s[j] = sv;
        cmax = fmaxf(cmax, sv);
      }
      const float mn = fmaxf(m, cmax);
      const float alpha = exp2f(m - mn);
      m = mn;
      l *= alpha;
#pragma unroll
      for (int d = 0; d < 64; d++) acc[d] *= alpha;
      static_assert(PA_CH == 16, "a chunk is one half of a keep-bit word");
      unsigned kbits = 0xFFFFu;
      if (DROP) kbits = brow[(k0 + c0) >> 5] >> ((k0 + c0) & 31);
#pragma unroll
      for (int j = 0; j < PA_CH; j++) {
        float p = valid_s[c0 + j] != 0.f ? exp2f(s[j] - mn) : 0.f;
        l += p;  // the normaliser is that of the undropped probabilities
        if (DROP) p = ((kbits >> j) & 1u) ? p * rkeep : 0.f;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(&Vs[c0 + j][d]);
          acc[d] = fmaf(p, vv.x, acc[d]); acc[d + 1] = fmaf(p, vv.y, acc[d + 1]);
          acc[d + 2] = fmaf(p, vv.z, acc[d + 2]); acc[d + 3] = fmaf(p, vv.w, acc[d + 3]);
        }
      }
    }
  }
  if (!qok) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const long oo = ((long)b * Np + qi) * ((long)H * 64) + h * 64;
#pragma unroll
  for (int d = 0; d < 64; d += 4) {
    const float a0 = acc[d] * inv, a1 = acc[d + 1] * inv, a2 = acc[d + 2] * inv, a3 = acc[d + 3] * inv;
    *reinterpret_cast<float4*>(o32 + oo + d) = make_float4(a0, a1, a2, a3);
    if (o16) *reinterpret_cast<uint2*>(o16 + oo + d) = make_uint2(pack_f16x2(a0, a1), pack_f16x2(a2, a3));
    if (ob) *reinterpret_cast<uint2*>(ob + oo + d) = make_uint2(pack_bf16x2(a0, a1), pack_bf16x2(a2, a3));
  }
  if (lse) lse[(long)bh * Np + qi] = m + log2f(l);
}

// GEGLU on the fp32 pre-activation in the packed column order (128-column blocks: 64 "x" then their 64 "gate" columns):
// g[r][64*t + j] = gelu_erf(h1[r][128*t + 64 + j]) * h1[r][128*t + j]    (voicebox_pytorch.py:338-340; libm erff, not the A&S form)
__global__ void geglu_f32_kernel(const float* __restrict__ h1, float* __restrict__ g32, u16* __restrict__ g16, u16* __restrict__ gb,
                                 u16* __restrict__ h1b, long M, int Fp) {
  const int cpr = Fp / 4;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cpr;
    const int f = (int)(i - r * cpr) * 4;
    const int t = f >> 6, j = f & 63;
    const long xo = r * 2 * Fp + 128L * t + j;
    const float4 xv = *reinterpret_cast<const float4*>(h1 + xo), gv = *reinterpret_cast<const float4*>(h1 + xo + 64);
    auto ge = [](float g, float x) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)) * x; };
    const float4 o = make_float4(ge(gv.x, xv.x), ge(gv.y, xv.y), ge(gv.z, xv.z), ge(gv.w, xv.w));
    const long go = r * Fp + f;
    *reinterpret_cast<float4*>(g32 + go) = o;
    if (g16) *reinterpret_cast<uint2*>(g16 + go) = make_uint2(pack_f16x2_sat(o.x, o.y), pack_f16x2_sat(o.z, o.w));
    if (gb) *reinterpret_cast<uint2*>(gb + go) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
    if (h1b) {
      *reinterpret_cast<uint2*>(h1b + xo) = make_uint2(pack_bf16x2(xv.x, xv.y), pack_bf16x2(xv.z, xv.w));
      *reinterpret_cast<uint2*>(h1b + xo + 64) = make_uint2(pack_bf16x2(gv.x, gv.y), pack_bf16x2(gv.z, gv.w));
    }
  }
}

// ada[j / group][b][j % group] = bias[j] + temb[b,:] . W[j,:]  with the fp32 master weights (voicebox_pytorch.py:273); one wave per (j, b)
__global__ __launch_bounds__(256) void adaln_f32_kernel(const float* __restrict__ temb, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ ada, int B, int Th, int J,
                                                        int group) {
  const int lane = threadIdx.x & 63;
  const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (long)J * B) return;
  const int j = (int)(item / B), b = (int)(item - (long)j * B);
  float s = 0.f;
  for (int t = lane * 4; t < Th; t += 256) {
    const float4 a = *reinterpret_cast<const float4*>(temb + (long)b * Th + t), ww = *reinterpret_cast<const float4*>(w + (long)j * Th + t);
    s = fmaf(a.x, ww.x, s); s = fmaf(a.y, ww.y, s); s = fmaf(a.z, ww.z, s); s = fmaf(a.w, ww.w, s);
  }
  s = wave_sum(s);
  if (lane == 0) ada[((long)(j / group) * B + b) * group + (j % group)] = s + bias[j];
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI (op level)
extern "C" int vbx_split3_f16(const float* src, long rows, int K, long ld, void* dst_f16, int Kp, void* stream) {
  VBX_REQUIRE(src && dst_f16 && rows > 0 && K > 0 && Kp >= K && Kp % 8 == 0 && ld % 4 == 0 && ld >= K, "vbx_split3_f16: bad args");
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * Kp / 4)), dim3(256), 0, ST, src, rows, K, ld, (u16*)dst_f16, Kp);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_pack_weight3(const float* src, int src_rows, int src_cols, void* dst_f16, int dst_rows, int dst_cols, int rowmap,
                                int F, void* stream) {
  VBX_REQUIRE(src && dst_f16 && dst_cols % 8 == 0, "vbx_pack_weight3: bad args");
  hipLaunchKernelGGL(pack_weight3_kernel, dim3(grid_for((long)dst_rows * dst_cols)), dim3(256), 0, ST, src, src_rows, src_cols,
                     (u16*)dst_f16, dst_rows, dst_cols, rowmap, F);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_qknorm_rope_f32(const float* raw, int B, int H, int Np, float qk_scale, const float* q_gamma, const float* k_gamma,
                                   const float* rot_cos, const float* rot_sin, float* q32, float* k32, float* v32, void* q16, void* k16,
                                   void* qb, void* kb, void* v_bf16, void* v16, float* q_rnorm, float* k_rnorm, float q16_scale,
                                   void* stream) {
  VBX_REQUIRE(raw && rot_cos && rot_sin && q32 && k32 && v32 && B > 0 && H > 0 && Np > 0, "vbx_qknorm_rope_f32: bad args");
  VBX_REQUIRE(qk_scale <= 0.f || (q_gamma && k_gamma), "vbx_qknorm_rope_f32: qk-norm needs gammas");
  const long M = (long)B * Np;
  hipLaunchKernelGGL(qknorm_rope_f32_kernel, dim3(cdiv(M * H, 4)), dim3(256), 0, ST, raw, Np, H, M, qk_scale, q_gamma, k_gamma, rot_cos,
                     rot_sin, q32, k32, v32, (u16*)q16, (u16*)k16, (u16*)qb, (u16*)kb, (u16*)v_bf16, (u16*)v16, q_rnorm, k_rnorm,
                     q16_scale > 0.f ? q16_scale : 1.0f);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_attn_fwd_f32_dropout(const float* q, const float* k, const float* v, const uint8_t* mask, float* out32, void* out16,
                                        void* out_bf16, float* lse, int B, int H, int Np, float scale, const void* bits_rm, float p,
                                        void* stream) {
  VBX_REQUIRE(q && k && v && out32 && bits_rm && B > 0 && H > 0 && Np > 0 && p > 0.f && p < 1.f, "vbx_attn_fwd_f32_dropout: bad args");
  hipLaunchKernelGGL(attn_fwd_f32_kernel<true>, dim3(cdiv(Np, 256), B * H), dim3(256), 0, ST, q, k, v, mask, out32, (u16*)out16, (u16*)out_bf16,
                     lse, H, Np, scale * 1.44269504088896340736f, (const unsigned*)bits_rm, vbx_dropout_bits_words(Np),
                     vbx_dropout_keep_scale(p));
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_attn_fwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask, float* out32, void* out16,
                                void* out_bf16, float* lse, int B, int H, int Np, float scale, void* stream) {
  VBX_REQUIRE(q && k && v && out32 && B > 0 && H > 0 && Np > 0, "vbx_attn_fwd_f32: bad args");
  hipLaunchKernelGGL(attn_fwd_f32_kernel<false>, dim3(cdiv(Np, 256), B * H), dim3(256), 0, ST, q, k, v, mask, out32, (u16*)out16,
                     (u16*)out_bf16, lse, H, Np, scale * 1.44269504088896340736f, (const unsigned*)nullptr, 0, 1.0f);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_geglu_f32(const float* h1, float* g32, void* g16, void* g_bf16, void* h1_bf16, long M, int Fp, void* stream) {
  VBX_REQUIRE(h1 && g32 && M > 0 && Fp > 0 && Fp % 64 == 0, "vbx_geglu_f32: bad args (Fp must be a multiple of 64)");
  hipLaunchKernelGGL(geglu_f32_kernel, dim3(grid_for(M * Fp / 4)), dim3(256), 0, ST, h1, g32, (u16*)g16, (u16*)g_bf16, (u16*)h1_bf16, M, Fp);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_adaln_proj_f32(const float* temb, const float* w, const float* bias, float* ada, int B, int Th, int J, int group,
                                  void* stream) {
  VBX_REQUIRE(temb && w && bias && ada && Th % 4 == 0 && B > 0 && J > 0, "vbx_adaln_proj_f32: bad args");
  if (group <= 0) group = J;
  VBX_REQUIRE(J % group == 0, "vbx_adaln_proj_f32: J must be a multiple of group");
  hipLaunchKernelGGL(adaln_f32_kernel, dim3(cdiv((long)J * B, 4)), dim3(256), 0, ST, temb, w, bias, ada, B, Th, J, group);
  VBX_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ stage level
namespace {
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
struct PCarver {
  char* base;
  size_t off = 0;
  explicit PCarver(void* b) : base((char*)b) {}
  template <class T>
  T* take(size_t n) {
    off = al256(off);
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};
struct PDims {
  int B, N, R, Np, D, H, I, F, Fp, Th, L, Din, E, Ke;
  long M, M0;
};
PDims pdims(const vbx_model* m) {
  PDims d;
  d.B = m->B; d.N = m->N; d.R = m->R; d.Np = m->N + m->R; d.D = m->D; d.H = m->H; d.I = m->H * 64;
  d.Din = m->Din > 0 ? m->Din : m->D;
  d.E = m->E;
  d.Ke = 2 * d.Din + d.E;  // to_embed input: [x | cond_emb | cond]  (:1075-1078)
  d.F = m->F; d.Fp = ((m->F + 63) / 64) * 64; d.Th = m->Th; d.L = m->L;
  d.M = (long)d.B * d.Np; d.M0 = (long)d.B * d.N;
  return d;
}
struct W3Layer { u16 *qkv, *out, *w1, *w2, *glw; };
struct W3 {
  u16 *emb, *pred;
  std::vector<W3Layer> layer;
  size_t bytes;
};
void carve_w3(const vbx_model* m, void* base, W3& w) {
  const PDims d = pdims(m);
  PCarver c(base);
  w.emb = c.take<u16>((size_t)d.D * 3 * d.Ke);
  w.pred = c.take<u16>((size_t)d.Din * 3 * d.D);
  w.layer.resize(d.L);
  for (int l = 0; l < d.L; l++) {
    w.layer[l].qkv = c.take<u16>((size_t)3 * d.I * 3 * d.D);
    w.layer[l].out = c.take<u16>((size_t)d.D * 3 * d.I);
    w.layer[l].w1 = c.take<u16>((size_t)2 * d.Fp * 3 * d.D);
    w.layer[l].w2 = c.take<u16>((size_t)d.D * 3 * d.Fp);
    w.layer[l].glw = m->gateloop ? c.take<u16>((size_t)3 * d.D * 3 * d.D) : nullptr;
  }
  w.bytes = al256(c.off);
}
struct PScratch {
  u16* a3;
  float *h32, *raw, *q32, *k32, *v32, *o32, *g32;
  size_t bytes;
};
void carve_ps(const vbx_model* m, void* base, PScratch& s) {
  const PDims d = pdims(m);
  PCarver c(base);
  const size_t kmax = (size_t)std::max(std::max(d.D, d.I), std::max(d.Fp, d.Ke));
  s.a3 = c.take<u16>((size_t)d.M * 3 * kmax);
  s.h32 = c.take<float>((size_t)d.M * std::max(d.D, d.Ke));
  s.raw = c.take<float>((size_t)d.M * std::max(3 * d.I, 2 * d.Fp));
  const size_t hs = (size_t)d.B * d.H * d.Np * 64;
  s.q32 = c.take<float>(hs);
  s.k32 = c.take<float>(hs);
  s.v32 = c.take<float>(hs);
  s.o32 = c.take<float>((size_t)d.M * d.I);
  s.g32 = c.take<float>((size_t)d.M * d.Fp);
  s.bytes = al256(c.off);
}
int supported(const vbx_model* m) {
  VBX_REQUIRE(!m->stack_only && !m->plain_norm && !m->unet,
              "precise mode serves the VoiceBox forward (not the standalone Transformer stack, u-net skips or plain RMSNorm)");
  return 0;
}
}  // namespace

extern "C" size_t vbx_model_precise_wpack_bytes(const vbx_model* m) {
  W3 w;
  carve_w3(m, nullptr, w);
  return w.bytes;
}
extern "C" size_t vbx_model_precise_scratch_bytes(const vbx_model* m) {
  PScratch s;
  carve_ps(m, nullptr, s);
  return s.bytes;
}
extern "C" int vbx_model_pack_weights_precise(const vbx_model* m, void* stream) {
  VBX_REQUIRE(m && m->params && m->off && m->wpack3, "vbx_model_pack_weights_precise: null field");
  if (int rc = supported(m)) return rc;
  const PDims d = pdims(m);
  W3 w;
  carve_w3(m, m->wpack3, w);
  const float* P = m->params;
  const long* G = m->off;
  if (int rc = vbx_pack_weight3(P + G[VBX_P_EMBW], d.D, d.Ke, w.emb, d.D, d.Ke, 0, 0, stream)) return rc;
  if (int rc = vbx_pack_weight3(P + G[VBX_P_PREDW], d.Din, d.D, w.pred, d.Din, d.D, 0, 0, stream)) return rc;
  for (int l = 0; l < d.L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    if (int rc = vbx_pack_weight3(P + o[VBX_L_QKVW], 3 * d.I, d.D, w.layer[l].qkv, 3 * d.I, d.D, 0, 0, stream)) return rc;
    if (int rc = vbx_pack_weight3(P + o[VBX_L_OUTW], d.D, d.I, w.layer[l].out, d.D, d.I, 0, 0, stream)) return rc;
    if (int rc = vbx_pack_weight3(P + o[VBX_L_FF1W], 2 * d.F, d.D, w.layer[l].w1, 2 * d.Fp, d.D, 1, d.F, stream)) return rc;
    if (int rc = vbx_pack_weight3(P + o[VBX_L_FF2W], d.D, d.F, w.layer[l].w2, d.D, d.Fp, 0, 0, stream)) return rc;
    if (m->gateloop)
      if (int rc = vbx_pack_weight3(P + o[VBX_L_GLW], 3 * d.D, d.D, w.layer[l].glw, 3 * d.D, d.D, 0, 0, stream)) return rc;
  }
  return 0;
}

// runtime.hip hands over the tensors of its own arenas that this forward fills for the (unchanged) backward
int vbx_forward_precise(const vbx_model* m, const vbx_io* io, const VbxPreciseActs* a, void* stream) {
  if (int rc = supported(m)) return rc;
  VBX_REQUIRE(m->wpack3 && m->pscratch, "vbx_model_forward: precise mode needs the wpack3 / pscratch arenas");
  const PDims d = pdims(m);
  W3 w;
  carve_w3(m, m->wpack3, w);
  PScratch s;
  carve_ps(m, m->pscratch, s);
  const float* P = m->params;
  const long* G = m->off;
  auto gemm3 = [&](const float* A32, long rows, int K, int Kp, const u16* W, int N, float* C, int ldc, const float* bias,
                   const float* resid) -> int {
    if (int rc = vbx_split3_f16(A32, rows, K, K, s.a3, Kp, stream)) return rc;
    vbx_gemm_desc g{};
    g.mode = VBX_GEMM_NT; g.epilogue = VBX_EPI_F32; g.M = (int)rows; g.N = N; g.K = 3 * Kp; g.lda = 3 * Kp; g.ldb = 3 * Kp; g.ldc = ldc;
    g.A = s.a3; g.B = W; g.C = C; g.bias = bias; g.resid = resid; g.f16 = 1;
    return vbx_gemm(&g, stream);
  };
#define PCK(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)
  // time embedding (fp32 already) + adaLN projections from the fp32 master weights   (:1082, :273)
  if (io->ada_table) {  // the sampler's precomputed table (vbx_model_adaln_table evaluates it with the fp32 weights in this mode)
    PCK(vbx_ada_select(a->ada, d.L, d.B, 4 * d.D, io->ada_table, io->ada_counter, io->ada_slot, stream));
  } else {
    PCK(vbx_time_embed_fwd(io->times, P + G[VBX_P_SINW], P + G[VBX_P_T1W], P + G[VBX_P_T1B], a->four, a->pre, a->temb, d.B, d.D, d.Th, stream));
    for (int l = 0; l < d.L; l++) {
      const long* o = m->off + VBX_NG + (long)l * VBX_NL;
      PCK(vbx_adaln_proj_f32(a->temb, P + o[VBX_L_G1W], P + o[VBX_L_G1B], a->ada + (size_t)l * d.B * 4 * d.D, d.B, d.Th, 4 * d.D, 4 * d.D, stream));
    }
  }
  // to_embed(cat(x, cond * ~cond_mask))   (:1035,1075-1078); the bf16 copy of the input is the backward's wgrad operand
  if (d.E) {  // condition_on_text (:1056-1078): [x | interpolated cond_emb rows | cond'] -- the same rows the fast path rounds to fp16
    VBX_REQUIRE(io->cond_ids && io->T > 0, "vbx_model_forward: a text-conditioned model needs cond_ids");
    if (a->embed_in_bf16)
      PCK(vbx_pack_embed_input_text(io->x, io->cond, io->cond_mask, io->drop_mask, io->null_cond, io->cond_ids, io->T, P + G[VBX_P_CEMB],
                                    d.E, io->null_id, a->embed_in_f16, a->embed_in_bf16, d.B, d.N, d.Din, stream));
    PCK(vbx_embed_input_text_f32(io->x, io->cond, io->cond_mask, io->drop_mask, io->null_cond, io->cond_ids, io->T, P + G[VBX_P_CEMB],
                                 d.E, io->null_id, s.h32, d.B, d.N, d.Din, stream));
  } else {
    if (a->embed_in_bf16)
      PCK(vbx_pack_embed_input(io->x, io->cond, io->cond_mask, a->embed_in_f16, a->embed_in_bf16, d.B, d.N, d.Din, stream));
    hipLaunchKernelGGL(embed_cat_kernel, dim3(grid_for(d.M0 * 2 * d.Din / 4)), dim3(256), 0, ST, io->x, io->cond, io->cond_mask, s.h32, d.M0, d.Din);
    VBX_LAUNCH_CHECK();
  }
  PCK(gemm3(s.h32, d.M0, d.Ke, d.Ke, w.emb, d.D, a->e, d.D, P + G[VBX_P_EMBB], nullptr));
  PCK(vbx_convpos_fwd_libm(a->e, P + G[VBX_P_CONVW], P + G[VBX_P_CONVB], io->attn_mask, d.R ? P + G[VBX_P_REG] : nullptr, a->xs[0], d.B, d.N, d.R, d.D, m->ksize, stream));
  for (int l = 0; l < d.L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    const VbxPreciseLayer& y = a->layer[l];
    const float* ada_l = a->ada + (size_t)l * d.B * 4 * d.D;
    const int S = m->gateloop ? 3 : 2;  // residual snapshots per layer (runtime.hip::carve_acts)
    float* x0 = a->xs[S * l];
    float* x_in = m->gateloop ? a->xs[S * l + 1] : x0;
    float* x_mid = a->xs[S * l + S - 1];
    float* x_out = a->xs[S * l + S];
    if (m->gateloop) {
      // x = GateLoop(x) + x   (:465-466): RMSNorm -> to_qkva (split GEMM) -> gated scan -> post LayerNorm + residual (fp32 on both paths)
      PCK(vbx_rmsnorm_fwd_multi(x0, P + o[VBX_L_GLG], nullptr, 0, y.hg, nullptr, s.h32, d.B, d.Np, 0, d.Np, d.D, stream));
      PCK(gemm3(s.h32, d.M, d.D, d.D, w.layer[l].glw, 3 * d.D, y.glp, 3 * d.D, nullptr, nullptr));
      PCK(vbx_gateloop_scan_fwd(y.glp, y.gls, y.glh, d.B, d.Np, d.D, stream));
      PCK(vbx_layernorm_fwd(y.gls, P + o[VBX_L_GLLNW], P + o[VBX_L_GLLNB], x0, x_in, d.M, d.D, 1e-5f, stream));
    }
    const bool drop_on = io->dropout != 0;
    PCK(vbx_rmsnorm_fwd_multi(x_in, ada_l, ada_l + d.D, 4 * d.D, y.hn1, nullptr, s.h32, d.B, d.Np, 0, d.Np, d.D, stream));
    PCK(gemm3(s.h32, d.M, d.D, d.D, w.layer[l].qkv, 3 * d.I, s.raw, 3 * d.I, nullptr, nullptr));
    PCK(vbx_qknorm_rope_f32(s.raw, d.B, d.H, d.Np, m->qk_norm ? 8.0f : 0.0f, m->qk_norm ? P + o[VBX_L_QG] : nullptr,
                            m->qk_norm ? P + o[VBX_L_KG] : nullptr, m->rot_cos, m->rot_sin, s.q32, s.k32, s.v32, y.q16, y.k16, y.qb, y.kb,
                            y.v, y.vh, y.qrn, y.krn, vbx_attn_q_prescale(m->attn_scale), stream));
    if (y.dbr && drop_on) {  // attend.py:131 with the fast path's keep bits (same seed and stream id: the backward re-reads them)
      PCK(vbx_attn_dropout_bits(y.dbr, y.dbc, d.B * d.H, d.Np, io->drop_seed, 2u * l, m->attn_dropout, stream));
      PCK(vbx_attn_fwd_f32_dropout(s.q32, s.k32, s.v32, io->attn_mask_p, s.o32, y.oh, y.o, y.lse, d.B, d.H, d.Np, m->attn_scale, y.dbr,
                                   m->attn_dropout, stream));
    } else {
      PCK(vbx_attn_fwd_f32(s.q32, s.k32, s.v32, io->attn_mask_p, s.o32, y.oh, y.o, y.lse, d.B, d.H, d.Np, m->attn_scale, stream));
    }
    PCK(gemm3(s.o32, d.M, d.I, d.I, w.layer[l].out, d.D, x_mid, d.D, nullptr, x_in));
    PCK(vbx_rmsnorm_fwd_multi(x_mid, ada_l + 2 * d.D, ada_l + 3 * d.D, 4 * d.D, y.hn2, nullptr, s.h32, d.B, d.Np, 0, d.Np, d.D, stream));
    PCK(gemm3(s.h32, d.M, d.D, d.D, w.layer[l].w1, 2 * d.Fp, s.raw, 2 * d.Fp, y.b1, nullptr));
    PCK(vbx_geglu_f32(s.raw, s.g32, y.gh, y.g, y.h1, d.M, d.Fp, stream));
    if (drop_on && m->ff_dropout > 0.f) {  // nn.Dropout between GEGLU and the output projection (:346): the mask of the fast path
      PCK(vbx_dropout_rows_f32(s.g32, d.M, d.Fp, d.Fp, io->drop_seed, 2u * l + 1u, m->ff_dropout, stream));
      PCK(vbx_dropout_rows(y.gh, y.g, d.M, d.Fp, d.Fp, io->drop_seed, 2u * l + 1u, m->ff_dropout, stream));
    }
    PCK(gemm3(s.g32, d.M, d.Fp, d.Fp, w.layer[l].w2, d.D, x_out, d.D, P + o[VBX_L_FF2B], x_mid));
  }
  // strip registers, final RMSNorm, to_pred, masked MSE   (:476-479, :1092, :1099-1115)
  PCK(vbx_rmsnorm_fwd_multi(a->xs[(m->gateloop ? 3 : 2) * d.L], P + G[VBX_P_FNG], nullptr, 0, a->hf, nullptr, s.h32, d.B, d.Np, d.R, d.N, d.D, stream));
  float* pred = io->pred ? io->pred : a->pred;
  PCK(gemm3(s.h32, d.M0, d.D, d.D, w.pred, d.Din, pred, d.Din, nullptr, nullptr));
  if (io->target) PCK(vbx_masked_mse_fwd(pred, io->target, io->loss_mask, a->per_b, io->loss, d.B, d.N, d.Din, stream));
#undef PCK
  return 0;
}
