// Memory-bound / small kernels of the VoiceBox hot path (everything that is not a big GEMM, a norm or
// attention): embed packing, conv positional embedding, time embedding, adaLN projections, GEGLU
// backward, column sums, masked MSE, CFM inputs, ODE axpy, weight packing, Adam, grad-norm.
#include "common.hpp"
#include "reduce_roles.hpp"
#include <stdlib.h>

namespace {

VBX_DEV void unpack8_bf16(const uint4 p, float v[8]) {
  const unsigned w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = bf16_to_f32((u16)(w[i] & 0xffff));
    v[2 * i + 1] = bf16_to_f32((u16)(w[i] >> 16));
  }
}
VBX_DEV void unpack8_f16(const uint4 p, float v[8]) {
  const unsigned w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = f16_to_f32((u16)(w[i] & 0xffff));
    v[2 * i + 1] = f16_to_f32((u16)(w[i] >> 16));
  }
}
VBX_DEV uint4 pack8_h(const float v[8]) {  // model inputs / activations: saturating (common.hpp)
  return make_uint4(pack_f16x2_sat(v[0], v[1]), pack_f16x2_sat(v[2], v[3]), pack_f16x2_sat(v[4], v[5]), pack_f16x2_sat(v[6], v[7]));
}
VBX_DEV uint4 pack8(const float v[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

// ---------------------------------------------------------------- embed input packing
__global__ void pack_embed_kernel(const float* __restrict__ x, const float* __restrict__ cond,
                                  const uint8_t* __restrict__ cmask, u16* __restrict__ out, u16* __restrict__ outb,
                                  long rows, int D) {
  const int cpr = D / 8;  // chunks per half row
  const long total = rows * 2 * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (2 * cpr);
    const int c = (int)(i - row * 2 * cpr);
    const bool second = c >= cpr;
    const int d = (second ? c - cpr : c) * 8;
    const float* src = (second ? cond : x) + row * D + d;
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (second && cmask && cmask[row]) {
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = 0.f;
    }
    const long oo = row * 2 * D + (second ? D : 0) + d;
    *reinterpret_cast<uint4*>(out + oo) = pack8_h(v);
    if (outb) *reinterpret_cast<uint4*>(outb + oo) = pack8(v);
  }
}

// ---- text-conditioned embed input (voicebox_pytorch.py:1035-1076): row = [ x | cond_emb | cond' ]
// cond' = where(drop[b], null_cond, cond * ~cond_mask)                                   (:1035, :1043-1048)
// cond_emb = to_cond_emb(where(drop[b], null_id, ids))  resized from T tokens to N frames (:1050-1066); the resize is
// F.interpolate(..., mode='bilinear', align_corners=False) over the frame axis (interpolate_1d, :89-107):
//   src = max(T/N * (n + 0.5) - 0.5, 0);  i0 = floor(src);  i1 = i0 + (i0 < T-1);  lam = src - i0;  (1-lam) e[i0] + lam e[i1]
VBX_DEV void interp_src(int n, int N, int T, int& i0, int& i1, float& lam) {
  if (T == N) { i0 = i1 = n; lam = 0.f; return; }
  const float scale = (float)T / (float)N;
  float src = scale * ((float)n + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  if (i0 > T - 1) i0 = T - 1;
  i1 = i0 + (i0 < T - 1 ? 1 : 0);
  lam = src - (float)i0;
}
__global__ void pack_embed_text_kernel(const float* __restrict__ x, const float* __restrict__ cond, const uint8_t* __restrict__ cmask,
                                       const uint8_t* __restrict__ drop, const float* __restrict__ null_cond,
                                       const long* __restrict__ ids, int T, const float* __restrict__ table, int E, long null_id,
                                       u16* __restrict__ out, u16* __restrict__ outb, float* __restrict__ out32, int B, int N, int D) {
  const int cx = D / 8, ce = E / 8, cpr = 2 * cx + ce;  // 8-wide chunks per output row
  const long total = (long)B * N * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    const int c = (int)(i - row * cpr);
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const bool dropped = drop && drop[b];
    float v[8];
    if (c < cx) {
      const float* src = x + row * D + c * 8;
      const float4 a = *reinterpret_cast<const float4*>(src), bb = *reinterpret_cast<const float4*>(src + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
    } else if (c < cx + ce) {
      const int e0 = (c - cx) * 8;
      int i0, i1;
      float lam;
      interp_src(n, N, T, i0, i1, lam);
      const long id0 = dropped ? null_id : ids[(long)b * T + i0], id1 = dropped ? null_id : ids[(long)b * T + i1];
      const float* r0 = table + id0 * E + e0;
      const float* r1 = table + id1 * E + e0;
      const float w0 = 1.0f - lam;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = (T == N) ? r0[k] : (w0 * r0[k] + lam * r1[k]);
    } else {
      const int d = (c - cx - ce) * 8;
      if (dropped) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = null_cond[d + k];
      } else if (cmask && cmask[row]) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = 0.f;
      } else {
        const float* src = cond + row * D + d;
        const float4 a = *reinterpret_cast<const float4*>(src), bb = *reinterpret_cast<const float4*>(src + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
      }
    }
    const long oo = row * (2L * D + E) + (long)c * 8;
    if (out) *reinterpret_cast<uint4*>(out + oo) = pack8_h(v);
    if (outb) *reinterpret_cast<uint4*>(outb + oo) = pack8(v);
    if (out32) {  // precise mode: the same rows, unrounded
      *reinterpret_cast<float4*>(out32 + oo) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out32 + oo + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}
// ---- DurationPredictor embed input (voicebox_pytorch.py:793-823): row (b, n) of N phoneme positions =
// [ to_phoneme_emb(max(ids, 0)) | cond'' ],  cond'' = curtail_or_pad(where(drop[b], null_cond, cond * ~cond_mask), N):
// frames n >= S are the zero padding of curtail_or_pad (:109-117), applied AFTER the drop (:797-804, :819).
__global__ void pack_phoneme_kernel(const long* __restrict__ ids, const float* __restrict__ table, int E,
                                    const float* __restrict__ cond, int S, const uint8_t* __restrict__ cmask,
                                    const uint8_t* __restrict__ drop, const float* __restrict__ null_cond,
                                    u16* __restrict__ out, int B, int N, int D) {
  const int ce = E / 8, cpr = ce + D / 8;
  const long total = (long)B * N * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    const int c = (int)(i - row * cpr);
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    float v[8];
    if (c < ce) {
      long id = ids[row];
      id = id < 0 ? 0 : id;  // -1 = padding, clamped (:811)
      const float* src = table + id * E + c * 8;
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = src[k];
    } else {
      const int d = (c - ce) * 8;
      const long crow = (long)b * S + n;
      if (n >= S || (!(drop && drop[b]) && cmask && cmask[crow])) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = 0.f;
      } else {
        const float* src = (drop && drop[b]) ? null_cond + d : cond + crow * D + d;
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = src[k];
      }
    }
    *reinterpret_cast<uint4*>(out + row * (long)(E + D) + (long)c * 8) = pack8_h(v);
  }
}
// to_pred = Linear(dim, 1) + Rearrange('... 1 -> ...') (:672-675): one wave per row.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out, long rows, int D) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + r * D + d), b = *reinterpret_cast<const float4*>(w + d);
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  s = wave_sum(s);
  if (lane == 0) out[r] = s + (bias ? bias[0] : 0.f);
}
// gradient of the embedding table: scatter of d(cond_emb) [B*N, E] (bf16) through the same resize weights.  fp32 atomics:
// like torch's embedding backward the accumulation order is not deterministic.
__global__ void cond_emb_bwd_kernel(const u16* __restrict__ demb, int ld, const long* __restrict__ ids, int T,
                                    const uint8_t* __restrict__ drop, long null_id, float* __restrict__ gtable, int B, int N, int E) {
  const long total = (long)B * N * E;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / E;
    const int e = (int)(i - row * E);
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const bool dropped = drop && drop[b];
    int i0, i1;
    float lam;
    interp_src(n, N, T, i0, i1, lam);
    const float g = bf16_to_f32(demb[row * ld + e]);
    const long id0 = dropped ? null_id : ids[(long)b * T + i0], id1 = dropped ? null_id : ids[(long)b * T + i1];
    if (T == N) {
      atomicAdd(gtable + id0 * E + e, g);
    } else {
      atomicAdd(gtable + id0 * E + e, (1.0f - lam) * g);
      atomicAdd(gtable + id1 * E + e, lam * g);
    }
  }
}

// ---------------------------------------------------------------- conv positional embedding
// tile: 64 frames x 64 channels per block; thread (dl = tid&63, ng = tid>>6) computes 16 frames of channel dl.
constexpr int CT = 64;
template <int MODE, int KS>  // MODE 0: forward -> xs (2: the same with libm's erff: precise mode) ; 1: dpre = dxs * m * gelu'(pre) -> out.  KS: compile-time kernel size
__global__ __launch_bounds__(256) void convpos_fwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ dxs, float* __restrict__ out, int N,
                                                          int R, int D) {
  extern __shared__ float tile[];  // [(CT + ks - 1)][64]
  constexpr int ks = KS;
  const int half = ks / 2;
  const int n0 = blockIdx.x * CT, dbase = blockIdx.y * 64, b = blockIdx.z;
  const int dl = threadIdx.x & 63, ng = threadIdx.x >> 6;
  const int d = dbase + dl;
  const int rows = CT + ks - 1;
  for (int r = ng; r < rows; r += 4) {
    const int n = n0 + r - half;
    float v = 0.f;
    if (n >= 0 && n < N && d < D && (!mask || mask[(long)b * N + n])) v = e[((long)b * N + n) * D + d];
    tile[r * 64 + dl] = v;
  }
  __syncthreads();
  if (d >= D) return;
  float wr[KS];
#pragma unroll
  for (int k = 0; k < ks; k++) wr[k] = w[(long)d * ks + k];
  const float bb = bias[d];
  const int Np = N + R;
  for (int i = 0; i < 16; i++) {
    const int nl = ng * 16 + i, n = n0 + nl;
    if (n >= N) break;
    float acc = bb;
#pragma unroll
    for (int k = 0; k < ks; k++) acc += wr[k] * tile[(nl + k) * 64 + dl];
    const bool m = !mask || mask[(long)b * N + n];
    if (MODE == 0 || MODE == 2) {
      // residual uses the UNMASKED e (voicebox_pytorch.py:1080 adds x, conv masks internally)
      const float ev = e[((long)b * N + n) * D + d];
      out[((long)b * Np + R + n) * D + d] = ev + (m ? (MODE == 2 ? gelu_erf_libm(acc) : gelu_erf(acc)) : 0.f);
    } else {
      const float g = dxs[((long)b * Np + R + n) * D + d];
      out[((long)b * N + n) * D + d] = m ? g * gelu_erf_grad(acc) : 0.f;
    }
  }
}

// standalone Transformer.forward (voicebox_pytorch.py:412-431): rows n >= R of the residual stream = the caller's x
__global__ void stack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int N, int R, int D, int to_stack) {
  // to_stack = 1: dst[b, R+n, :] = src[b, n, :] ([B,N,D] -> [B,N+R,D]);  0: dst[b, n, :] = src[b, R+n, :]
  const long total = (long)B * N * D / 4;
  const int D4 = D / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / D4;
    const int c = (int)(i - row * D4);
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const long wide = ((long)b * (N + R) + R + n) * D4 + c, narrow = i;
    if (to_stack) reinterpret_cast<float4*>(dst)[wide] = reinterpret_cast<const float4*>(src)[narrow];
    else reinterpret_cast<float4*>(dst)[narrow] = reinterpret_cast<const float4*>(src)[wide];
  }
}

// u-net skip connections of the standalone Transformer (voicebox_pytorch.py:458-463)
__global__ void unet_cat_kernel(const float* __restrict__ x, const float* __restrict__ skip, float scale, u16* __restrict__ o16,
                                u16* __restrict__ ob, long rows, int D) {
  const int cpr = D / 4;
  const long total = rows * 2 * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / (2 * cpr);
    const int c = (int)(i - row * 2 * cpr);
    const bool second = c >= cpr;
    const int d = (second ? c - cpr : c) * 4;
    float4 v = *reinterpret_cast<const float4*>((second ? skip : x) + row * D + d);
    if (second) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
    const long oo = row * 2 * D + (second ? D : 0) + d;
    if (o16) *reinterpret_cast<uint2*>(o16 + oo) = make_uint2(pack_f16x2_sat(v.x, v.y), pack_f16x2_sat(v.z, v.w));
    if (ob) *reinterpret_cast<uint2*>(ob + oo) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
}
__global__ void unet_split_kernel(const float* __restrict__ dcat, float scale, float* __restrict__ dx, u16* __restrict__ dxb,
                                  float* __restrict__ dskip, long rows, int D) {
  const int cpr = D / 4;
  const long total = rows * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cpr;
    const int d = (int)(i - row * cpr) * 4;
    const float4 a = *reinterpret_cast<const float4*>(dcat + row * 2 * D + d);
    float4 b = *reinterpret_cast<const float4*>(dcat + row * 2 * D + D + d);
    b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
    *reinterpret_cast<float4*>(dx + row * D + d) = a;
    if (dxb) *reinterpret_cast<uint2*>(dxb + row * D + d) = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
    *reinterpret_cast<float4*>(dskip + row * D + d) = b;
  }
}
__global__ void unet_addskip_kernel(float* __restrict__ dx, u16* __restrict__ dxb, const float* __restrict__ dskip, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(dx)[i];
    const float4 b = reinterpret_cast<const float4*>(dskip)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4*>(dx)[i] = a;
    if (dxb) reinterpret_cast<uint2*>(dxb)[i] = make_uint2(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w));
  }
}

__global__ void regs_fill_kernel(const float* __restrict__ reg, float* __restrict__ xs, int B, int Np, int R, int D) {
  const long total = (long)B * R * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)R * D));
    const long rd = i - (long)b * R * D;
    xs[(long)b * Np * D + rd] = reg[rd];
  }
}

// de[b,n,d] = dxs[b,R+n,d] + m[b,n] * sum_k w[d][k] dpre[b, n-k+half, d]
// wpart[chunk][d][k] = sum_{n in tile} dpre[b,n,d] * em[b, n+k-half, d] (k<ks) ; wpart[chunk][d][63] = sum dpre
template <int KS>
__global__ __launch_bounds__(256) void convpos_bwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                          const uint8_t* __restrict__ mask, const float* __restrict__ dxs,
                                                          const float* __restrict__ dpre, float* __restrict__ de,
                                                          u16* __restrict__ deb, float* __restrict__ wpart, int N, int R,
                                                          int D) {
  extern __shared__ float sm[];  // e tile [rows][64] | dpre tile [rows][64]; afterwards the reduction buffer [4][64][33]
  constexpr int ks = KS;
  static_assert(KS <= 31, "wacc holds the taps in [0, KS) and the bias gradient in [31]");
  const int half = ks / 2;
  const int rows = CT + ks - 1;
  float* te = sm;
  float* tp = sm + rows * 64;
  const int n0 = blockIdx.x * CT, dbase = blockIdx.y * 64, b = blockIdx.z;
  const int dl = threadIdx.x & 63, ng = threadIdx.x >> 6;
  const int d = dbase + dl;
  for (int r = ng; r < rows; r += 4) {
    const int n = n0 + r - half;
    float ve = 0.f, vp = 0.f;
    if (n >= 0 && n < N && d < D) {
      if (!mask || mask[(long)b * N + n]) ve = e[((long)b * N + n) * D + d];
      vp = dpre[((long)b * N + n) * D + d];
    }
    te[r * 64 + dl] = ve;
    tp[r * 64 + dl] = vp;
  }
  __syncthreads();
  float wr[KS], wacc[32];
#pragma unroll
  for (int k = 0; k < 32; k++) wacc[k] = 0.f;
  if (d < D) {
#pragma unroll
    for (int k = 0; k < ks; k++) wr[k] = w[(long)d * ks + k];
    const int Np = N + R;
    for (int i = 0; i < 16; i++) {
      const int nl = ng * 16 + i, n = n0 + nl;
      if (n >= N) break;
      // out position nl <-> tile row nl + half
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < ks; k++) acc += wr[k] * tp[(nl + 2 * half - k) * 64 + dl];  // dpre[n - k + half]
      const bool m = !mask || mask[(long)b * N + n];
      const float o = dxs[((long)b * Np + R + n) * D + d] + (m ? acc : 0.f);
      de[((long)b * N + n) * D + d] = o;
      if (deb) deb[((long)b * N + n) * D + d] = f32_to_bf16(o);
      const float dp = tp[(nl + half) * 64 + dl];
#pragma unroll
      for (int k = 0; k < ks; k++) wacc[k] += dp * te[(nl + k) * 64 + dl];
      wacc[31] += dp;
    }
  }
  // The tiles are dead once every thread has left the loop: their space becomes the cross-group reduction buffer
  // (48 KiB of LDS per block instead of 112 KiB -> three blocks per CU).  Row pitch 33: conflict-free for the per-tap writes
  // (lanes = channels) and for the reads below (lanes = taps), which make the partial-record stores 128-byte contiguous
  // (they were 4-byte stores at a 256-byte stride).  Entries [KS, 62] of a record are never read (conv_wgrad_finalize).
  __syncthreads();
  float* red = sm;
#pragma unroll
  for (int k = 0; k < 32; k++) red[(ng * 64 + dl) * 33 + k] = wacc[k];
  __syncthreads();
  const long chunk = (long)b * gridDim.x + blockIdx.x;
  for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
    const int c = idx >> 5, k = idx & 31;
    if (dbase + c < D) {
      const float s = red[(0 * 64 + c) * 33 + k] + red[(1 * 64 + c) * 33 + k] + red[(2 * 64 + c) * 33 + k] +
                      red[(3 * 64 + c) * 33 + k];
      wpart[(chunk * D + dbase + c) * 64 + (k == 31 ? 63 : k)] = s;
    }
  }
}

// dw[d][k] = sum_chunk wpart[chunk][d][k] (k < ks) ; db[d] = sum_chunk wpart[chunk][d][63]
__global__ void conv_wgrad_finalize_kernel(const float* __restrict__ wpart, int chunks, int D, int ks, float* __restrict__ dw,
                                           float* __restrict__ db) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= D * 64) return;
  const int d = idx >> 6, k = idx & 63;
  if (k >= ks && k != 63) return;
  float s = 0.f;
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {  // four records in flight (same summation order)
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = wpart[((long)(c + u) * D + d) * 64 + k];
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u];
  }
  for (; c < chunks; c++) s += wpart[((long)c * D + d) * 64 + k];
  if (k == 63) db[d] = s;
  else dw[(long)d * ks + k] = s;
}

__global__ void dreg_kernel(const float* __restrict__ dxs, float* __restrict__ dreg, int B, int Np, int R, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * D) return;
  float s = 0.f;
  for (int b = 0; b < B; b++) s += dxs[(long)b * Np * D + i];
  dreg[i] = s;
}

// ---------------------------------------------------------------- time embedding
__global__ void time_four_kernel(const float* __restrict__ times, const float* __restrict__ wsin, float* __restrict__ four,
                                 int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, j = i - b * D, hd = D / 2;
  const int jj = j < hd ? j : j - hd;
  float f = times[b] * wsin[jj];  // x * weights * 2 * pi  (voicebox_pytorch.py:165)
  f = f * 2.0f;
  f = f * 3.14159265358979323846f;
  four[i] = j < hd ? sinf(f) : cosf(f);
}
// one wave per (b, i): pre = b1[i] + four[b,:] . W1[i,:]
__global__ __launch_bounds__(256) void time_linear_kernel(const float* __restrict__ four, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, float* __restrict__ pre,
                                                           float* __restrict__ temb, int B, int D, int Th) {
  const int lane = threadIdx.x & 63;
  const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long)B * Th) return;
  const int b = (int)(o / Th), i = (int)(o - (long)b * Th);
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += four[(long)b * D + j] * w1[(long)i * D + j];
  s = wave_sum(s);
  if (lane == 0) {
    const float p = s + b1[i];
    pre[o] = p;
    temb[o] = p / (1.0f + expf(-p));
  }
}
VBX_DEV float silu_grad(float p) {
  const float sg = 1.0f / (1.0f + expf(-p));
  return sg * (1.0f + p * (1.0f - sg));
}
__global__ void time_bwd_w1_kernel(const float* __restrict__ four, const float* __restrict__ pre,
                                   const float* __restrict__ dtemb, float* __restrict__ dw1, float* __restrict__ db1, int B,
                                   int D, int Th) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)Th * D) return;
  const int i = (int)(idx / D), j = (int)(idx - (long)i * D);
  float s = 0.f, sb = 0.f;
  for (int b = 0; b < B; b++) {
    const float dp = dtemb[(long)b * Th + i] * silu_grad(pre[(long)b * Th + i]);
    s += dp * four[(long)b * D + j];
    sb += dp;
  }
  dw1[idx] = s;
  if (j == 0) db1[i] = sb;
}
// partial[slice][b][j] = sum_{i in slice} dtemb[b,i] silu'(pre[b,i]) W1[i,j]; grid (D/64, B, TB_SLICES), 64 threads
constexpr int TB_SLICES = 32;
__global__ void time_bwd_four_kernel(const float* __restrict__ w1, const float* __restrict__ pre,
                                     const float* __restrict__ dtemb, float* __restrict__ partial, int B, int D, int Th) {
  const int j = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y, sl = blockIdx.z;
  if (j >= D) return;
  const int per = (Th + TB_SLICES - 1) / TB_SLICES;
  const int ib = sl * per, ie = min(Th, ib + per);
  float s = 0.f;
  for (int i = ib; i < ie; i++) s += dtemb[(long)b * Th + i] * silu_grad(pre[(long)b * Th + i]) * w1[(long)i * D + j];
  partial[((long)sl * B + b) * D + j] = s;
}
__global__ void time_bwd_wsin_kernel(const float* __restrict__ times, const float* __restrict__ four,
                                     const float* __restrict__ dfour, float* __restrict__ dwsin, int B, int D) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, hd = D / 2;
  if (j >= hd) return;
  float s = 0.f;
  for (int b = 0; b < B; b++) {
    const float tw = times[b] * 2.0f * 3.14159265358979323846f;
    const float sn = four[(long)b * D + j], cs = four[(long)b * D + hd + j];
    s += tw * (dfour[(long)b * D + j] * cs - dfour[(long)b * D + hd + j] * sn);
  }
  dwsin[j] = s;
}

// ---------------------------------------------------------------- adaLN projections (weight streaming, M = B)
// block: 256 threads = 4 waves, 64 outputs (16 per wave); temb staged in LDS [bc][Th] fp32, bc <= 8.
__global__ __launch_bounds__(256) void adaln_fwd_kernel(const float* __restrict__ temb, const u16* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ ada, int B,
                                                        int Th, int J, int bc, int group) {
  extern __shared__ __attribute__((aligned(16))) float st[];  // [bc][Th]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < B; b0 += bc) {
    const int nb = min(bc, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * Th; i += 256) st[i] = temb[(long)b0 * Th + i];
    __syncthreads();
    for (int jj = 0; jj < 16; jj++) {
      const int j = blockIdx.x * 64 + wave * 16 + jj;
      if (j >= J) break;
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; k++) acc[k] = 0.f;
      for (int c = lane; c < Th / 8; c += 64) {
        float wv[8];
        unpack8_f16(*reinterpret_cast<const uint4*>(w + (long)j * Th + c * 8), wv);
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (k < nb) {
            const float4 t0 = *reinterpret_cast<const float4*>(st + k * Th + c * 8);
            const float4 t1 = *reinterpret_cast<const float4*>(st + k * Th + c * 8 + 4);
            acc[k] += wv[0] * t0.x + wv[1] * t0.y + wv[2] * t0.z + wv[3] * t0.w + wv[4] * t1.x + wv[5] * t1.y +
                      wv[6] * t1.z + wv[7] * t1.w;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (k < nb) {
          const float s = wave_sum(acc[k]);
          if (lane == 0) ada[((long)(j / group) * B + (b0 + k)) * group + (j % group)] = s + bias[j];
        }
      }
    }
  }
}

// adaLN projections as a register-level MFMA "GEMV": ada[b][j] = bias[j] + temb[b,:] . W[j,:]   (B <= 16, W fp16 [J,Th])
// One wave owns 16 outputs j and streams their weight rows straight from HBM into MFMA B fragments (k contiguous: no
// LDS); temb (fp32, L2 resident) becomes the A fragment as an fp16 hi + lo pair (two MFMAs) so the time conditioning keeps
// ~22 mantissa bits.  Weight-streaming bound: 100 MB at dim 512 / depth 12.
__global__ __launch_bounds__(256) void adaln_fwd_mfma_kernel(const float* __restrict__ temb, const u16* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ ada, int B,
                                                             int Th, int J, int group) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + wave) * 16;
  if (j0 >= J) return;
  const int jr = min(j0 + (lane & 15), J - 1), kq = (lane >> 4) * 8, bi = lane & 15;
  const u16* wrow = w + (long)jr * Th + kq;
  const float* trow = temb + (long)min(bi, B - 1) * Th + kq;
  const bool bval = bi < B;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;  // k-steps (32 wide) kept in flight
  for (int k0 = 0; k0 < Th; k0 += 32 * U) {
    uint4 wf[U];
    float4 t0[U], t1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + 32 * u;
      if (k < Th) {
        wf[u] = *reinterpret_cast<const uint4*>(wrow + k);
        t0[u] = *reinterpret_cast<const float4*>(trow + k);
        t1[u] = *reinterpret_cast<const float4*>(trow + k + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int k = k0 + 32 * u;
      if (k < Th) {
        const float tv[8] = {t0[u].x, t0[u].y, t0[u].z, t0[u].w, t1[u].x, t1[u].y, t1[u].z, t1[u].w};
        f16x8 ahi, alo;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float x = bval ? tv[i] : 0.f;
          const _Float16 h = (_Float16)x;
          ahi[i] = h;
          alo[i] = (_Float16)(x - (float)h);
        }
        const f16x8 bf = __builtin_bit_cast(f16x8, wf[u]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bf, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bf, acc, 0, 0, 0);
      }
    }
  }
  // C layout: col j = lane&15, row b = (lane>>4)*4 + r
  const int j = j0 + (lane & 15);
  if (j < J) {
    const float bj = bias[j];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int b = (lane >> 4) * 4 + r;
      if (b < B) ada[((long)(j / group) * B + b) * group + (j % group)] = acc[r] + bj;
    }
  }
}

// dW[j][t] = sum_b dada[b][j] temb[b][t] ; dbias[j] = sum_b dada[b][j]
// a block writes ADA_WROWS consecutive rows j: the B float4 of temb a thread needs are loaded once and reused per row
constexpr int ADA_WROWS = 4;
VBX_DEV void adaln_bwd_w_role(const float* __restrict__ temb, const float* __restrict__ dada, float* __restrict__ dw,
                                 float* __restrict__ dbias, int B, int Th, int J, int jblk) {
  const int t4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (t4 * 4 >= Th) return;
  const int j0 = jblk * ADA_WROWS;
  if (B <= 8) {
    float4 t[8];
#pragma unroll
    for (int b = 0; b < 8; b++)
      t[b] = b < B ? *reinterpret_cast<const float4*>(temb + (long)b * Th + t4 * 4) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < ADA_WROWS; r++) {
      const int j = j0 + r;
      if (j >= J) break;
      float4 s = make_float4(0, 0, 0, 0);
      float sb = 0.f;
#pragma unroll
      for (int b = 0; b < 8; b++) {
        const float g = b < B ? dada[(long)b * J + j] : 0.f;
        s.x += g * t[b].x; s.y += g * t[b].y; s.z += g * t[b].z; s.w += g * t[b].w;
        sb += g;
      }
      if (dw) *reinterpret_cast<float4*>(dw + (long)j * Th + t4 * 4) = s;
      if (t4 == 0) dbias[j] = sb;
    }
    return;
  }
  for (int r = 0; r < ADA_WROWS; r++) {
    const int j = j0 + r;
    if (j >= J) break;
    float4 s = make_float4(0, 0, 0, 0);
    float sb = 0.f;
    for (int b = 0; b < B; b++) {
      const float g = dada[(long)b * J + j];
      const float4 t = *reinterpret_cast<const float4*>(temb + (long)b * Th + t4 * 4);
      s.x += g * t.x; s.y += g * t.y; s.z += g * t.z; s.w += g * t.w;
      sb += g;
    }
    if (dw) *reinterpret_cast<float4*>(dw + (long)j * Th + t4 * 4) = s;
    if (t4 == 0) dbias[j] = sb;
  }
}
// partial dtemb over a slice of j: scratch[slice][b][t]
constexpr int ADA_SLICES = 128;
VBX_DEV void adaln_bwd_t_role(const u16* __restrict__ w, const float* __restrict__ dada, float* __restrict__ scratch, int B, int Th,
                                 int J, int slice) {
  const int per = (J + ADA_SLICES - 1) / ADA_SLICES;
  const int jb = slice * per, je = min(J, jb + per);
  const int t8 = blockIdx.x * blockDim.x + threadIdx.x;  // chunk of 8 t
  if (t8 * 8 >= Th) return;
  for (int b0 = 0; b0 < B; b0 += 8) {
    const int nb = min(8, B - b0);
    float acc[8][8];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
#pragma unroll 4
    for (int j = jb; j < je; j++) {
      float wv[8];
      unpack8_f16(*reinterpret_cast<const uint4*>(w + (long)j * Th + t8 * 8), wv);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (k < nb) {
          const float g = dada[(long)(b0 + k) * J + j];
#pragma unroll
          for (int i = 0; i < 8; i++) acc[k][i] += g * wv[i];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (k < nb) {
        float* o = scratch + ((long)slice * B + b0 + k) * Th + t8 * 8;
        *reinterpret_cast<float4*>(o) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
      }
    }
  }
}
// one launch for both halves of the adaLN projection backward: blockIdx.y < ADA_SLICES -> the d(time_emb) partial of that slice,
// otherwise the weight/bias gradient of output rows [ADA_WROWS * (blockIdx.y - ADA_SLICES), +ADA_WROWS)
__global__ __launch_bounds__(256) void adaln_bwd_kernel(const float* __restrict__ temb, const u16* __restrict__ w,
                                                        const float* __restrict__ dada, float* __restrict__ dw,
                                                        float* __restrict__ dbias, float* __restrict__ scratch, int B, int Th, int J) {
  // the few long-running d(time_emb) blocks (a serial loop over a slice of j) are dispatched first so that they overlap the
  // write-bound weight-gradient blocks instead of forming the kernel's tail
  if ((int)blockIdx.y < ADA_SLICES) adaln_bwd_t_role(w, dada, scratch, B, Th, J, blockIdx.y);
  else adaln_bwd_w_role(temb, dada, dw, dbias, B, Th, J, blockIdx.y - ADA_SLICES);
}

// v2 of the same launch (default; VBX_ADALN_BWD=1 selects the kernel above for the A/B): the compiler turned v1's per-batch
// guards into scalar branches with a wait after every load (one exposed memory round trip per row of the slice -- the
// d(time_emb) blocks were the kernel's critical path at ~20 us).  Here the block's dada values sit in LDS, batch rows
// beyond B carry a zero weight instead of a branch, and ADA_TG weight rows are requested before the first is consumed.
constexpr int ADA_PER_MAX = 64;  // rows of one d(time_emb) slice held in LDS: J <= ADA_SLICES * 64
constexpr int ADA_TG = 8;        // weight rows requested together by a d(time_emb) thread
__global__ __launch_bounds__(256) void adaln_bwd_kernel_v2(const float* __restrict__ temb, const u16* __restrict__ w,
                                                           const float* __restrict__ dada, float* __restrict__ dw,
                                                           float* __restrict__ dbias, float* __restrict__ scratch, int B, int Th,
                                                           int J) {
  __shared__ __attribute__((aligned(16))) float gsh[ADA_PER_MAX * 8];  // [row of the block][8 batch rows]
  const int tid = threadIdx.x;
  if ((int)blockIdx.y < ADA_SLICES) {  // ---- d(time_emb) partial of one slice of j: scratch[slice][b][t]
    if ((long)blockIdx.x * 256 * 8 >= Th) return;  // block-uniform
    const int slice = blockIdx.y;
    const int per = (J + ADA_SLICES - 1) / ADA_SLICES;
    const int jb = slice * per, je = min(J, jb + per);
    const int t8 = blockIdx.x * 256 + tid;
    const bool tv = (long)t8 * 8 < Th;
    const u16* wcol = w + (tv ? t8 : 0) * 8;
    for (int b0 = 0; b0 < B; b0 += 8) {
      __syncthreads();
      for (int i = tid; i < per * 8; i += 256) {
        const int j = jb + (i >> 3), b = b0 + (i & 7);
        gsh[i] = (j < je && b < B) ? dada[(long)b * J + j] : 0.f;
      }
      __syncthreads();
      float acc[8][8];
#pragma unroll
      for (int k = 0; k < 8; k++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
      for (int jj0 = 0; jj0 < per; jj0 += ADA_TG) {
        uint4 wq[ADA_TG];
#pragma unroll
        for (int u = 0; u < ADA_TG; u++) wq[u] = *reinterpret_cast<const uint4*>(wcol + (long)min(jb + jj0 + u, J - 1) * Th);
#pragma unroll
        for (int u = 0; u < ADA_TG; u++) {
          if (jj0 + u < per) {  // rows past the slice were clamped above and carry no weight
            float wv[8];
            unpack8_f16(wq[u], wv);
            const float4 g0 = *reinterpret_cast<const float4*>(gsh + (jj0 + u) * 8);
            const float4 g1 = *reinterpret_cast<const float4*>(gsh + (jj0 + u) * 8 + 4);
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
              for (int i = 0; i < 8; i++) acc[k][i] += gv[k] * wv[i];
          }
        }
      }
      if (tv) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
          if (b0 + k < B) {
            float* o = scratch + ((long)slice * B + b0 + k) * Th + (long)t8 * 8;
            *reinterpret_cast<float4*>(o) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
          }
        }
      }
    }
    return;
  }
  // ---- weight / bias gradient of rows [j0, j0 + ADA_WROWS)
  const int j0 = ((int)blockIdx.y - ADA_SLICES) * ADA_WROWS;
  const int t4 = blockIdx.x * 256 + tid;
  const bool tv = (long)t4 * 4 < Th;
  const float* tcol = temb + (tv ? t4 : 0) * 4;
  float4 s[ADA_WROWS];
  float sb[ADA_WROWS];
#pragma unroll
  for (int r = 0; r < ADA_WROWS; r++) { s[r] = make_float4(0.f, 0.f, 0.f, 0.f); sb[r] = 0.f; }
  for (int b0 = 0; b0 < B; b0 += 8) {
    __syncthreads();
    if (tid < 8 * ADA_WROWS) {
      const int j = j0 + (tid >> 3), b = b0 + (tid & 7);
      gsh[tid] = (j < J && b < B) ? dada[(long)b * J + j] : 0.f;
    }
    __syncthreads();
    float4 t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = *reinterpret_cast<const float4*>(tcol + (long)min(b0 + k, B - 1) * Th);  // weight 0 past B
#pragma unroll
    for (int r = 0; r < ADA_WROWS; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float g = gsh[r * 8 + k];
        s[r].x += g * t[k].x; s[r].y += g * t[k].y; s[r].z += g * t[k].z; s[r].w += g * t[k].w;
        sb[r] += g;
      }
    }
  }
  if (tv) {
#pragma unroll
    for (int r = 0; r < ADA_WROWS; r++) {
      const int j = j0 + r;
      if (j < J) {
        if (dw) *reinterpret_cast<float4*>(dw + (long)j * Th + (long)t4 * 4) = s[r];  // NULL: the gradient stays in factor form
        if (t4 == 0) dbias[j] = sb[r];
      }
    }
  }
}

// ---- factor form (vbx_model.adaln_factors): the weight gradient is never formed and the bias gradient is a column sum of the norm
// partial records, so what is left of the adaLN backward is d(time_emb)[b][t] = sum over (layer, j) of dada[l][b][j] * W[l][j][t].
// ONE launch per step for every layer (per layer it was a latency-bound 11 us kernel + a 4.7 us reduction, x depth): slices of
// ADA_PER_MAX rows of the stacked [L * J, Th] weight -> scratch[slice][b][t], then sum_rows_wide_kernel.
__global__ __launch_bounds__(256) void adaln_dtemb_all_kernel(const u16* __restrict__ w, const float* __restrict__ dada,
                                                              float* __restrict__ scratch, int B, int Th, int J, int spl) {
  __shared__ __attribute__((aligned(16))) float gsh[ADA_PER_MAX * 8];  // [row of the slice][8 batch rows]
  const int tid = threadIdx.x;
  if ((long)blockIdx.x * 256 * 8 >= Th) return;  // block-uniform
  const int slice = blockIdx.y, l = slice / spl;
  const int jb = (slice - l * spl) * ADA_PER_MAX, je = min(J, jb + ADA_PER_MAX), per = je - jb;
  const float* dl = dada + (long)l * B * J;      // dada [L][B][J]
  const u16* wl = w + (long)l * J * Th;          // W [L][J][Th] fp16
  const int t8 = blockIdx.x * 256 + tid;
  const bool tv = (long)t8 * 8 < Th;
  const u16* wcol = wl + (tv ? t8 : 0) * 8;
  for (int b0 = 0; b0 < B; b0 += 8) {
    __syncthreads();
    for (int i = tid; i < ADA_PER_MAX * 8; i += 256) {
      const int j = jb + (i >> 3), b = b0 + (i & 7);
      gsh[i] = (j < je && b < B) ? dl[(long)b * J + j] : 0.f;
    }
    __syncthreads();
    float acc[8][8];
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[k][i] = 0.f;
    for (int jj0 = 0; jj0 < per; jj0 += ADA_TG) {
      uint4 wq[ADA_TG];
#pragma unroll
      for (int u = 0; u < ADA_TG; u++) wq[u] = *reinterpret_cast<const uint4*>(wcol + (long)min(jb + jj0 + u, J - 1) * Th);
#pragma unroll
      for (int u = 0; u < ADA_TG; u++) {  // rows past the slice were clamped above and carry a zero weight in gsh
        float wv[8];
        unpack8_f16(wq[u], wv);
        const float4 g0 = *reinterpret_cast<const float4*>(gsh + (jj0 + u) * 8);
        const float4 g1 = *reinterpret_cast<const float4*>(gsh + (jj0 + u) * 8 + 4);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int i = 0; i < 8; i++) acc[k][i] += gv[k] * wv[i];
      }
    }
    if (tv) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (b0 + k < B) {
          float* o = scratch + ((long)slice * B + b0 + k) * Th + (long)t8 * 8;
          *reinterpret_cast<float4*>(o) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
          *reinterpret_cast<float4*>(o + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
        }
      }
    }
  }
}
// out[j] = sum_i in[i*ld + j] for MANY rows: block = 32 columns x 32 row lanes, four rows in flight per lane
__global__ __launch_bounds__(1024) void sum_rows_wide_kernel(const float* __restrict__ in, long rows, long ld, float* __restrict__ out,
                                                             long cols) {
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long j = blockIdx.x * 32L + cl;
  float s = 0.f;
  if (j < cols) {
    long i = rl;
    for (; i + 96 < rows; i += 128) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = in[(i + 32 * u) * ld + j];
#pragma unroll
      for (int u = 0; u < 4; u++) s += v[u];
    }
    for (; i < rows; i += 32) s += in[i * ld + j];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && j < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; k++) t += red[k][cl];
    out[j] = t;
  }
}

// out[j] (+)= sum_i in[i*ld + j]
// block = 64 columns x 4 row lanes (launch with 256 threads, grid = cdiv(cols, 64))
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ in, long rows, long ld, float* __restrict__ out,
                                                        long cols, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long j = blockIdx.x * 64L + cl;
  float s = 0.f;
  if (j < cols) {
    long i = rl;
    for (; i + 12 < rows; i += 16) {  // four rows in flight (same summation order)
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = in[(i + 4 * u) * ld + j];
#pragma unroll
      for (int u = 0; u < 4; u++) s += v[u];
    }
    for (; i < rows; i += 4) s += in[i * ld + j];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && j < cols) {
    const float t = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    out[j] = accumulate ? out[j] + t : t;
  }
}

// ---------------------------------------------------------------- GEGLU backward (interleaved layout)
__global__ void geglu_bwd_kernel(const u16* __restrict__ h1, const u16* __restrict__ dg, u16* __restrict__ dh1, long M,
                                 int Fp) {
  const int cpr = Fp / 8;
  const long total = M * cpr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cpr;
    const int f = (int)(i - r * cpr) * 8;  // column in [0, Fp)
    const int blk = f >> 6, c = f & 63;
    const long xo = r * 2 * Fp + blk * 128 + c, go = xo + 64;
    float xv[8], gv[8], dv[8], dx[8], dgt[8];
    unpack8_bf16(*reinterpret_cast<const uint4*>(h1 + xo), xv);
    unpack8_bf16(*reinterpret_cast<const uint4*>(h1 + go), gv);
    unpack8_bf16(*reinterpret_cast<const uint4*>(dg + r * Fp + f), dv);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      dx[k] = dv[k] * gelu_erf(gv[k]);
      dgt[k] = dv[k] * xv[k] * gelu_erf_grad(gv[k]);
    }
    *reinterpret_cast<uint4*>(dh1 + xo) = pack8(dx);
    *reinterpret_cast<uint4*>(dh1 + go) = pack8(dgt);
  }
}

// ---------------------------------------------------------------- column sums
// stage 1: block = 64 column groups (16 B each) x 4 row lanes over one of CS_SLABS row slabs -> scratch[slab][C]
constexpr int CS_SLABS = 128;
// geglu_bwd that also emits the column sums of dh1 (FeedForward[0].bias gradient) as per-slab partial records, so the
// separate 47 MB colsum pass disappears.  grid (ceil(Fp/8/64), GB_SLABS); block = 64 eight-feature groups x 4 row lanes.
constexpr int GB_SLABS = 512;
__global__ __launch_bounds__(256) void geglu_bwd_colsum_kernel(const u16* __restrict__ h1, const u16* __restrict__ dg,
                                                               u16* __restrict__ dh1, long M, int Fp, float* __restrict__ scratch) {
  __shared__ float red[4][64][16];
  const int cgi = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int f = (blockIdx.x * 64 + cgi) * 8;  // feature in [0, Fp)
  const int slab = blockIdx.y;
  const long per = (M + GB_SLABS - 1) / GB_SLABS;
  const long rb = slab * per, re = min(M, rb + per);
  float ax[8], ag[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { ax[k] = 0.f; ag[k] = 0.f; }
  const int blk = f >> 6, c = f & 63;
  if (f < Fp) {
    for (long r = rb + rl; r < re; r += 4) {
      const long xo = r * 2 * Fp + blk * 128 + c, go = xo + 64;
      float xv[8], gv[8], dv[8], dx[8], dgt[8];
      unpack8_bf16(*reinterpret_cast<const uint4*>(h1 + xo), xv);
      unpack8_bf16(*reinterpret_cast<const uint4*>(h1 + go), gv);
      unpack8_bf16(*reinterpret_cast<const uint4*>(dg + r * Fp + f), dv);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        dx[k] = dv[k] * gelu_erf(gv[k]);
        dgt[k] = dv[k] * xv[k] * gelu_erf_grad(gv[k]);
      }
      const uint4 px = pack8(dx), pg = pack8(dgt);
      *reinterpret_cast<uint4*>(dh1 + xo) = px;
      *reinterpret_cast<uint4*>(dh1 + go) = pg;
      float rx[8], rg[8];  // the sums are over the STORED (bf16) values, like the weight gradient that reads dh1
      unpack8_bf16(px, rx);
      unpack8_bf16(pg, rg);
#pragma unroll
      for (int k = 0; k < 8; k++) { ax[k] += rx[k]; ag[k] += rg[k]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) { red[rl][cgi][k] = ax[k]; red[rl][cgi][8 + k] = ag[k]; }
  __syncthreads();
  if (rl == 0 && f < Fp) {
    float* o = scratch + (long)slab * 2 * Fp + blk * 128 + c;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      o[k] = red[0][cgi][k] + red[1][cgi][k] + red[2][cgi][k] + red[3][cgi][k];
      o[64 + k] = red[0][cgi][8 + k] + red[1][cgi][8 + k] + red[2][cgi][8 + k] + red[3][cgi][8 + k];
    }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void colsum_stage1(const void* __restrict__ in, long M, int C, long ld, float* __restrict__ scratch) {
  constexpr int W = BF16 ? 8 : 4;  // columns per 16-byte load
  __shared__ float red[4][64][W];
  const int cgi = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c0 = (blockIdx.x * 64 + cgi) * W;
  const int slab = blockIdx.y;
  const long per = (M + CS_SLABS - 1) / CS_SLABS;
  const long rb = slab * per, re = min(M, rb + per);
  float acc[W];
#pragma unroll
  for (int i = 0; i < W; i++) acc[i] = 0.f;
  if (c0 < C) {
    for (long r = rb + rl; r < re; r += 4) {
      if (BF16) {
        float v[8];
        unpack8_bf16(*reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(in) + r * ld + c0), v);
#pragma unroll
        for (int i = 0; i < W; i++) acc[i] += v[i % 8];
      } else {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + r * ld + c0);
        acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < W; i++) red[rl][cgi][i] = acc[i];
  __syncthreads();
  if (rl == 0 && c0 < C) {
#pragma unroll
    for (int i = 0; i < W; i++)
      scratch[(long)slab * C + c0 + i] = red[0][cgi][i] + red[1][cgi][i] + red[2][cgi][i] + red[3][cgi][i];
  }
}
// ---- batched small reductions: out[b][map(c)] = sum_{r < rows} src[b*src_bstride + r*row_stride + c] for up to VBX_MR_MAX
// independent jobs in ONE launch.  The backward of a layer ends with seven such reductions of partial records (norm gamma/beta,
// bias column sums, qk-norm gammas); as separate launches they cost ~0.5 ms of a 12.7 ms step -- almost all of it launch /
// drain latency (measured by skipping them).  Block = 64 columns x 16 row lanes.
__global__ __launch_bounds__(1024) void multi_reduce_kernel(vbx_mr_jobs jobs) {
  __shared__ float red[16][64];
  int j = 0;
#pragma unroll
  for (int i = 1; i < VBX_MR_MAX; i++)
    if (i < jobs.n && (int)blockIdx.x >= jobs.job[i].block0) j = i;
  const vbx_mr_job jb = jobs.job[j];
  mr_role(jb, blockIdx.x - jb.block0, red);  // reduce_roles.hpp
}
// Both job tables of a layer's backward in one launch: blocks [0, skr_blocks) reduce weight-gradient slabs (1024 threads x 4 columns
// each), the rest run the small column reductions.  Results are bit-identical to the two separate launches (same per-element order).
__global__ __launch_bounds__(1024) void layer_reduce_kernel(vbx_skr_jobs sj, vbx_mr_jobs mj, int skr_blocks) {
  __shared__ float red[16][64];
  if ((int)blockIdx.x < skr_blocks) {  // block-uniform
    int j = 0;
#pragma unroll
    for (int i = 1; i < VBX_SKR_MAX; i++)
      if (i < sj.n && (int)blockIdx.x >= sj.job[i].block0) j = i;
    const vbx_skr_job jb = sj.job[j];
    (void)skr_role(jb, ((long)(blockIdx.x - jb.block0) * 1024 + threadIdx.x) * 4);  // (no gradient-norm partials here: sq must be NULL)
    return;
  }
  const int bx = blockIdx.x - skr_blocks;
  int j = 0;
#pragma unroll
  for (int i = 1; i < VBX_MR_MAX; i++)
    if (i < mj.n && bx >= mj.job[i].block0) j = i;
  const vbx_mr_job jb = mj.job[j];
  mr_role(jb, bx - jb.block0, red);
}
__global__ void colsum_stage2(const float* __restrict__ scratch, int C, float* __restrict__ out, int out_len, int rowmap,
                              int F) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  int dc = c;
  if (rowmap == 1) dc = geglu_row_unmap(c, F);
  if (dc < 0 || dc >= out_len) return;
  float s = 0.f;
  static_assert(CS_SLABS % 4 == 0, "slab loop is unrolled by four");
  for (int k = 0; k < CS_SLABS; k += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = scratch[(long)(k + u) * C + c];
#pragma unroll
    for (int u = 0; u < 4; u++) s += v[u];
  }
  out[dc] = s;
}

// ---------------------------------------------------------------- masked MSE
// per_b[b] = sum_n mask * mean_d (p-t)^2 / max(count,1e-5) ; per_b[B+b] = den
constexpr int MSE_SPLITS = 64;  // 64 x B workgroups: at B = 8 every CU gets two (16 splits left half the chip idle: 31 us for a 34 MB read)
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       const uint8_t* __restrict__ lmask, float* __restrict__ per_b, int B,
                                                       int N, int D) {
  __shared__ float red[8];
  const int b = blockIdx.y, sp = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f, cnt = 0.f;
  for (int n = sp * 4 + wave; n < N; n += 4 * MSE_SPLITS) {
    if (!lmask[(long)b * N + n]) continue;
    const float4* p4 = reinterpret_cast<const float4*>(pred + ((long)b * N + n) * D);
    const float4* t4 = reinterpret_cast<const float4*>(target + ((long)b * N + n) * D);
    float s = 0.f;
    for (int c = lane; c < D / 4; c += 64) {
      const float4 p = p4[c], t = t4[c];
      const float a0 = p.x - t.x, a1 = p.y - t.y, a2 = p.z - t.z, a3 = p.w - t.w;
      s += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
    s = wave_sum(s);
    acc += s / (float)D;
    cnt += 1.f;
  }
  if (lane == 0) { red[wave] = acc; red[4 + wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* part = per_b + 2 * B + ((long)b * MSE_SPLITS + sp) * 2;
    part[0] = red[0] + red[1] + red[2] + red[3];
    part[1] = red[4] + red[5] + red[6] + red[7];
  }
}
// one wave: lane k owns split k of every batch element (MSE_SPLITS == 64); fixed butterfly order -> deterministic
__global__ void mse_mean_kernel(float* __restrict__ per_b, float* __restrict__ loss, int B) {
  static_assert(MSE_SPLITS == 64, "one split per lane");
  const int lane = threadIdx.x;
  float s = 0.f;
  for (int b = 0; b < B; b++) {
    const float2 pc = *reinterpret_cast<const float2*>(per_b + 2 * B + ((long)b * MSE_SPLITS + lane) * 2);
    const float num = wave_sum(pc.x), cnt = wave_sum(pc.y);
    const float den = fmaxf(cnt, 1e-5f);
    if (lane == 0) {
      per_b[b] = num / den;
      per_b[B + b] = den;
    }
    s += num / den;
  }
  if (lane == 0) loss[0] = s / (float)B;
}
// dpred = gscale * 2 (p-t) / D * mask / (den[b] * B)
__global__ void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                               const uint8_t* __restrict__ lmask, const float* __restrict__ per_b,
                               const float* __restrict__ gscale, float* __restrict__ dpred, u16* __restrict__ dpb, int B,
                               int N, int D) {
  const long total = (long)B * N * D / 4;
  const float gs = gscale ? gscale[0] : 1.0f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = (i * 4) / D;
    const int b = (int)(row / N);
    float4 o = make_float4(0, 0, 0, 0);
    if (lmask[row]) {
      const float k = gs * 2.0f / ((float)D * per_b[B + b] * (float)B);
      const float4 p = reinterpret_cast<const float4*>(pred)[i], t = reinterpret_cast<const float4*>(target)[i];
      o = make_float4(k * (p.x - t.x), k * (p.y - t.y), k * (p.z - t.z), k * (p.w - t.w));
    }
    if (dpred) reinterpret_cast<float4*>(dpred)[i] = o;
    if (dpb) reinterpret_cast<uint2*>(dpb)[i] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
  }
}

// ---------------------------------------------------------------- CFM inputs / ODE axpy
__global__ void cfm_inputs_kernel(const float* __restrict__ x1, const float* __restrict__ x0, const float* __restrict__ times,
                                  float sigma, float* __restrict__ w, float* __restrict__ flow, int B, long per) {
  const long total = (long)B * per;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float t = times[i / per];
    const float a = x0[i], c = x1[i];
    // w = (1 - (1 - sigma) t) x0 + t x1 ; flow = x1 - (1 - sigma) x0     (voicebox_pytorch.py:1408,1410)
    w[i] = (1.0f - (1.0f - sigma) * t) * a + t * c;
    flow[i] = c - (1.0f - sigma) * a;
  }
}
__global__ void axpy_dev_kernel(const float* __restrict__ y, const float* __restrict__ f, const float* __restrict__ coef,
                                int idx, float* __restrict__ out, long n4) {
  const float a = coef[idx];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 yy = reinterpret_cast<const float4*>(y)[i], ff = reinterpret_cast<const float4*>(f)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(yy.x + ff.x * a, yy.y + ff.y * a, yy.z + ff.z * a, yy.w + ff.w * a);
  }
}

// device-counter variants for the hipGraph-captured ODE step: the captured kernels read t / dt from
// device tables indexed by a device counter, so one captured step replays for every interval.
__global__ void ode_set_time_kernel(float* __restrict__ times, int B, const float* __restrict__ table,
                                    const int* __restrict__ counter, int slot) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) times[b] = table[2 * counter[0] + slot];
}
// ada[l][b][:] = table[2 * counter + slot][l][:] for every b: the adaLN projections of one ODE time point, precomputed for the whole
// grid (every batch element of a sampling call shares the time, so the table has no batch axis); G = floats per layer (4 * D)
__global__ void ada_select_kernel(float* __restrict__ ada, int L, int B, int G, const float* __restrict__ table,
                                  const int* __restrict__ counter, int slot) {
  const long n4 = (long)L * G / 4;
  const float4* src = reinterpret_cast<const float4*>(table + (long)(2 * counter[0] + slot) * L * G);
  const int g4 = G / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    const long l = i / g4, c = i - l * g4;
    for (int b = 0; b < B; b++) reinterpret_cast<float4*>(ada + ((l * B + b) * (long)G))[c] = v;
  }
}
__global__ void axpy_ctr_kernel(const float* __restrict__ y, const float* __restrict__ f, const float* __restrict__ table,
                                const int* __restrict__ counter, int slot, float* __restrict__ out, long n4) {
  const float a = table[2 * counter[0] + slot];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 yy = reinterpret_cast<const float4*>(y)[i], ff = reinterpret_cast<const float4*>(f)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(yy.x + ff.x * a, yy.y + ff.y * a, yy.z + ff.z * a, yy.w + ff.w * a);
  }
}
__global__ void counter_add_kernel(int* counter, int inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += inc;
}
// one idle wave for `ticks` x 10 ns: start-phase offset between the sampler's two half-batch streams (solver.py)
__global__ void stream_delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ---------------------------------------------------------------- weight packing
__global__ void pack_weight_kernel(const float* __restrict__ src, int src_rows, int src_cols, u16* __restrict__ dst,
                                   u16* __restrict__ dst16, int dst_rows, int dst_cols, int rowmap, int F) {
  const long total = (long)dst_rows * dst_cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i / dst_cols), c = (int)(i - (long)p * dst_cols);
    int r = p;
    if (rowmap == 1) r = geglu_row_unmap(p, F);
    float v = 0.f;
    if (r >= 0 && r < src_rows && c < src_cols) v = src[(long)r * src_cols + c];
    if (dst) dst[i] = f32_to_bf16(v);
    if (dst16) dst16[i] = f32_to_f16(v);
  }
}
__global__ void pack_bias_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int dst_n, int rowmap, int F) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= dst_n) return;
  int r = p;
  if (rowmap == 1) r = geglu_row_unmap(p, F);
  dst[p] = (r >= 0 && r < n) ? src[r] : 0.f;
}

// ---------------------------------------------------------------- optimizer
// Adam + packed-copy refresh (multi-tensor apply over a segment table; one block = 2048 consecutive floats of one segment)
constexpr int ADAM_BLOCK = 2048;
struct AdamCoef { float lr_bc1, b1, b2, eps, bc2_sqrt, gs; };
VBX_DEV float adam_one(float pv, float gv, float& mv, float& vv, const AdamCoef& k) {
  const float gr = gv * k.gs;
  mv = k.b1 * mv + (1.0f - k.b1) * gr;
  vv = k.b2 * vv + (1.0f - k.b2) * gr * gr;
  // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  return pv - k.lr_bc1 * mv / (sqrtf(vv) / k.bc2_sqrt + k.eps);
}
__global__ __launch_bounds__(256) void adam_packed_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const vbx_adam_seg* __restrict__ segs, int nsegs,
                                                          float lr_bc1, float b1, float b2, float eps, float bc2_sqrt,
                                                          const float* __restrict__ gscale) {
  // segment of this block: last s with segs[s].block0 <= blockIdx.x
  int lo = 0, hi = nsegs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].block0 <= (long)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const vbx_adam_seg sg = segs[lo];
  const AdamCoef k{lr_bc1, b1, b2, eps, bc2_sqrt, gscale ? gscale[0] : 1.0f};
  const long j0 = ((long)blockIdx.x - sg.block0) * ADAM_BLOCK;
  u16* db = (u16*)sg.dst_bf16;
  u16* dh = (u16*)sg.dst_f16;
  const bool packed = db || dh || sg.dst_f32;
  // 4 consecutive floats per lane (16-byte accesses, 8-byte packed stores) whenever a group cannot straddle a row
  const bool vec = (sg.off & 3) == 0 && (!packed || ((sg.cols & 3) == 0 && (sg.dst_ld & 3) == 0));
#pragma unroll
  for (int it = 0; it < ADAM_BLOCK / 1024; it++) {
    const long j = j0 + (long)(it * 256 + threadIdx.x) * 4;
    if (j >= sg.count) break;
    const long i = sg.off + j;
    if (vec && j + 3 < sg.count) {
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 mv = *reinterpret_cast<const float4*>(m + i), vv = *reinterpret_cast<const float4*>(v + i);
      float4 pv = *reinterpret_cast<const float4*>(p + i);
      pv.x = adam_one(pv.x, gv.x, mv.x, vv.x, k);
      pv.y = adam_one(pv.y, gv.y, mv.y, vv.y, k);
      pv.z = adam_one(pv.z, gv.z, mv.z, vv.z, k);
      pv.w = adam_one(pv.w, gv.w, mv.w, vv.w, k);
      *reinterpret_cast<float4*>(m + i) = mv;
      *reinterpret_cast<float4*>(v + i) = vv;
      *reinterpret_cast<float4*>(p + i) = pv;
      if (packed) {
        const int r = (int)(j / sg.cols), c = (int)(j - (long)r * sg.cols);
        int dr = r;
        if (sg.rowmap == 1) dr = r < sg.F ? ((r >> 6) << 7) + (r & 63) : ((((r - sg.F) >> 6) << 7) + 64 + ((r - sg.F) & 63));
        const long o = (long)dr * sg.dst_ld + c;
        if (db) *reinterpret_cast<uint2*>(db + o) = make_uint2(pack_bf16x2(pv.x, pv.y), pack_bf16x2(pv.z, pv.w));
        if (dh) *reinterpret_cast<uint2*>(dh + o) = make_uint2(pack_f16x2(pv.x, pv.y), pack_f16x2(pv.z, pv.w));
        if (sg.dst_f32) *reinterpret_cast<float4*>(sg.dst_f32 + o) = pv;
      }
    } else {
      for (int e = 0; e < 4 && j + e < sg.count; e++) {
        float mv = m[i + e], vv = v[i + e];
        const float pn = adam_one(p[i + e], g[i + e], mv, vv, k);
        m[i + e] = mv;
        v[i + e] = vv;
        p[i + e] = pn;
        if (packed) {
          const long jj = j + e;
          const int r = (int)(jj / sg.cols), c = (int)(jj - (long)r * sg.cols);
          int dr = r;
          if (sg.rowmap == 1) dr = r < sg.F ? ((r >> 6) << 7) + (r & 63) : ((((r - sg.F) >> 6) << 7) + 64 + ((r - sg.F) & 63));
          const long o = (long)dr * sg.dst_ld + c;
          if (db) db[o] = f32_to_bf16(pn);
          if (dh) dh[o] = f32_to_f16(pn);
          if (sg.dst_f32) sg.dst_f32[o] = pn;
        }
      }
    }
  }
}
// ---- adaLN projection weights in FACTOR form (round 5) ------------------------------------------------------------------------
// The gradient of the adaLN projection weight block of a layer, W_l [J4 = 4 D rows, Th cols], is the rank-B outer product
//   dW_l = dada_l^T . temb        (dada_l [B, J4]: d(gamma | beta) of the layer's two norms per batch row, temb [B, Th])
// -- 49 % of all parameters at dim 512 / depth 12, written (201 MB), re-read by the norm pass and re-read by Adam every step although
// it is defined by B * (J4 + Th) numbers.  In factor mode (vbx_model.adaln_factors) the backward never writes it: Adam expands the
// product on the fly, the global norm takes its sum of squares from two B x B Gram matrices per layer, and a data-parallel run puts
// the factors on the wire instead of the product (dp.py).
// Adam: one block = ADAF_ROWS weight rows x 1024 columns; a thread keeps its 4 columns of every temb row in registers (8 batch rows
// per pass) and walks the rows: the factors cost 4 bytes of L2 traffic per parameter instead of 4 bytes of HBM read + the 8 bytes
// the materialised gradient cost elsewhere.
constexpr int ADAF_ROWS = 8;  // (16: 128 block-uniform dada scalars per pass -> 534 spilled SGPRs, 3.8 TB/s; 8: 64 scalars)
struct AdaFactorArgs {
  long w_off[32];     // flat offset of W_l (floats)
  u16* dst_f16[32];   // packed fp16 copy of W_l ([J4][Th], row stride Th)
  const float* dada;  // [L][B][J4]
  const float* temb;  // [B][Th]
  int L, B, J4, Th;
};
__global__ __launch_bounds__(256) void adam_adaln_factor_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                const AdaFactorArgs a, float lr_bc1, float b1, float b2, float eps,
                                                                float bc2_sqrt, const float* __restrict__ gscale) {
  const int l = blockIdx.z, r0 = blockIdx.y * ADAF_ROWS;
  const int c_raw = (blockIdx.x * 256 + threadIdx.x) * 4;
  const bool cv = c_raw < a.Th;       // threads past the row stay in the block (barriers below) with a clamped column and no stores
  const int c = cv ? c_raw : 0;
  const AdamCoef k{lr_bc1, b1, b2, eps, bc2_sqrt, gscale ? gscale[0] : 1.0f};
  const float* da = a.dada + (long)l * a.B * a.J4;
  const long base = a.w_off[l] + (long)r0 * a.Th + c;
  const int nrows = min(ADAF_ROWS, a.J4 - r0);
  // the optimizer state of the block's first rows is requested BEFORE the gradient is expanded (their latency hides under the FMAs)
  constexpr int G = 4;  // rows whose p / m / v are in flight together
  float4 pv[G], mv[G], vv[G];
#pragma unroll
  for (int r = 0; r < G; r++) {
    const long i = base + (long)min(r, nrows - 1) * a.Th;
    pv[r] = *reinterpret_cast<const float4*>(p + i); mv[r] = *reinterpret_cast<const float4*>(m + i); vv[r] = *reinterpret_cast<const float4*>(v + i);
  }
  float4 acc[ADAF_ROWS];
#pragma unroll
  for (int r = 0; r < ADAF_ROWS; r++) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  __shared__ float gsh[ADAF_ROWS * 8];  // the block's dada values of one pass (rows x 8 batch rows), read back as LDS broadcasts
  for (int b0 = 0; b0 < a.B; b0 += 8) {
    __syncthreads();
    if (threadIdx.x < ADAF_ROWS * 8) {
      const int j = min(r0 + ((int)threadIdx.x >> 3), a.J4 - 1), b = b0 + ((int)threadIdx.x & 7);
      gsh[threadIdx.x] = b < a.B ? da[(long)b * a.J4 + j] : 0.f;
    }
    __syncthreads();
    float4 t[8];
#pragma unroll
    for (int kk = 0; kk < 8; kk++)
      t[kk] = *reinterpret_cast<const float4*>(a.temb + (long)min(b0 + kk, a.B - 1) * a.Th + c);  // weight 0 past B
#pragma unroll
    for (int r = 0; r < ADAF_ROWS; r++) {
#pragma unroll
      for (int kk = 0; kk < 8; kk++) {
        const float g = gsh[r * 8 + kk];
        acc[r].x += g * t[kk].x; acc[r].y += g * t[kk].y; acc[r].z += g * t[kk].z; acc[r].w += g * t[kk].w;
      }
    }
  }
  u16* dh = a.dst_f16[l];
#pragma unroll
  for (int g0 = 0; g0 < ADAF_ROWS; g0 += G) {
    float4 pn[G], mn[G], vn[G];
    if (g0 + G < ADAF_ROWS) {  // next group's state: requested before this group is computed and stored
#pragma unroll
      for (int r = 0; r < G; r++) {
        const long i = base + (long)min(g0 + G + r, nrows - 1) * a.Th;
        pn[r] = *reinterpret_cast<const float4*>(p + i); mn[r] = *reinterpret_cast<const float4*>(m + i); vn[r] = *reinterpret_cast<const float4*>(v + i);
      }
    }
#pragma unroll
    for (int r = 0; r < G; r++) {
      if (cv && g0 + r < nrows) {
        const long i = base + (long)(g0 + r) * a.Th;
        const float4 gq = acc[g0 + r];
        pv[r].x = adam_one(pv[r].x, gq.x, mv[r].x, vv[r].x, k);
        pv[r].y = adam_one(pv[r].y, gq.y, mv[r].y, vv[r].y, k);
        pv[r].z = adam_one(pv[r].z, gq.z, mv[r].z, vv[r].z, k);
        pv[r].w = adam_one(pv[r].w, gq.w, mv[r].w, vv[r].w, k);
        *reinterpret_cast<float4*>(m + i) = mv[r];
        *reinterpret_cast<float4*>(v + i) = vv[r];
        *reinterpret_cast<float4*>(p + i) = pv[r];
        if (dh) *reinterpret_cast<uint2*>(dh + (long)(r0 + g0 + r) * a.Th + c) = make_uint2(pack_f16x2(pv[r].x, pv[r].y), pack_f16x2(pv[r].z, pv[r].w));
      }
    }
    if (g0 + G < ADAF_ROWS) {
#pragma unroll
      for (int r = 0; r < G; r++) { pv[r] = pn[r]; mv[r] = mn[r]; vv[r] = vn[r]; }
    }
  }
}
// dW [J4, Th] = dada^T . temb for ANY number of batch rows B (the data-parallel exchange gathers every rank's factors and expands the
// summed gradient locally: dp.GradBucketReducer(adaln="factors")) -- the weight-gradient half of adaln_bwd_kernel_v2 on its own.
__global__ __launch_bounds__(256) void adaln_expand_dw_kernel(const float* __restrict__ temb, const float* __restrict__ dada,
                                                              float* __restrict__ dw, int B, int Th, int J) {
  __shared__ float gsh[8 * ADA_WROWS];
  const int tid = threadIdx.x;
  const int j0 = (int)blockIdx.y * ADA_WROWS;
  const int t4 = blockIdx.x * 256 + tid;
  const bool tv = (long)t4 * 4 < Th;
  const float* tcol = temb + (tv ? t4 : 0) * 4;
  float4 s[ADA_WROWS];
#pragma unroll
  for (int r = 0; r < ADA_WROWS; r++) s[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < B; b0 += 8) {
    __syncthreads();
    if (tid < 8 * ADA_WROWS) {
      const int j = j0 + (tid >> 3), b = b0 + (tid & 7);
      gsh[tid] = (j < J && b < B) ? dada[(long)b * J + j] : 0.f;
    }
    __syncthreads();
    float4 t[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t[k] = *reinterpret_cast<const float4*>(tcol + (long)min(b0 + k, B - 1) * Th);  // weight 0 past B
#pragma unroll
    for (int r = 0; r < ADA_WROWS; r++) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float g = gsh[r * 8 + k];
        s[r].x += g * t[k].x; s[r].y += g * t[k].y; s[r].z += g * t[k].z; s[r].w += g * t[k].w;
      }
    }
  }
  if (tv) {
#pragma unroll
    for (int r = 0; r < ADA_WROWS; r++)
      if (j0 + r < J) *reinterpret_cast<float4*>(dw + (long)(j0 + r) * Th + (long)t4 * 4) = s[r];
  }
}
// |dada_l^T . temb|_F^2 = sum_{b,b'} (dada_l[b] . dada_l[b']) (temb[b] . temb[b'])  ->  out[l * B * B + b * B + b'] (one term per
// block: 12 x 64 blocks at the benchmark shape; the first version -- one block per layer walking its 36 pairs -- took 114 us)
__global__ __launch_bounds__(256) void adaln_factor_sumsq_kernel(const float* __restrict__ dada, const float* __restrict__ temb,
                                                                 float* __restrict__ out, int B, int J4, int Th) {
  __shared__ float red[8];
  const int l = blockIdx.y, b = blockIdx.x / B, bp = blockIdx.x - b * B;
  const float* da = dada + (long)l * B * J4;
  float g1 = 0.f, g2 = 0.f;
  for (int j = threadIdx.x; j < J4; j += 256) g1 += da[(long)b * J4 + j] * da[(long)bp * J4 + j];
  for (int t = threadIdx.x; t < Th; t += 256) g2 += temb[(long)b * Th + t] * temb[(long)bp * Th + t];
  g1 = wave_sum(g1);
  g2 = wave_sum(g2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave] = g1; red[4 + wave] = g2; }
  __syncthreads();
  if (threadIdx.x == 0)
    out[((long)l * B + b) * B + bp] = (red[0] + red[1] + red[2] + red[3]) * (red[4] + red[5] + red[6] + red[7]);
}
// sum of squares over up to 32 ranges [lo, hi) of a flat buffer (every lo / hi a multiple of 4 floats, 16-byte aligned base):
// the flat gradient buffer minus the adaLN weight blocks that stay in factor form
struct SumsqRanges { long lo[64], pre[65]; int n; };  // pre[k] = float4 count of ranges 0..k-1
__global__ __launch_bounds__(256) void sumsq_ranges_stage1(const float* __restrict__ x, const SumsqRanges rg, float* __restrict__ scratch) {
  // the virtual concatenation of the ranges is cut into gridDim.x contiguous chunks; the table sits in LDS (indexing a kernel-argument
  // array by a run-time value made the first version walk scratch memory: 79 us for 209 MB)
  __shared__ long slo[64], spre[65];
  __shared__ float red[4];
  if (threadIdx.x < 64) slo[threadIdx.x] = rg.lo[threadIdx.x];
  if (threadIdx.x < 65) spre[threadIdx.x] = rg.pre[threadIdx.x];
  __syncthreads();
  const int n = rg.n;
  const long n4 = spre[n];
  const long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long c0 = blockIdx.x * per, c1 = min(n4, c0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int kr = 0;
  while (kr + 1 < n && c0 >= spre[kr + 1]) kr++;
  long i = c0 + threadIdx.x;
  while (i < c1) {
    while (i >= spre[kr + 1]) kr++;  // the index only grows
    const long seg_end = min(c1, spre[kr + 1]);
    const float4* base = reinterpret_cast<const float4*>(x + slo[kr]) - spre[kr];
    for (; i + 768 < seg_end; i += 1024) {  // four independent 16-byte loads in flight per lane
      const float4 v0 = base[i], v1 = base[i + 256], v2 = base[i + 512], v3 = base[i + 768];
      s0 = fmaf(v0.x, v0.x, s0); s1 = fmaf(v0.y, v0.y, s1); s2 = fmaf(v0.z, v0.z, s2); s3 = fmaf(v0.w, v0.w, s3);
      s0 = fmaf(v1.x, v1.x, s0); s1 = fmaf(v1.y, v1.y, s1); s2 = fmaf(v1.z, v1.z, s2); s3 = fmaf(v1.w, v1.w, s3);
      s0 = fmaf(v2.x, v2.x, s0); s1 = fmaf(v2.y, v2.y, s1); s2 = fmaf(v2.z, v2.z, s2); s3 = fmaf(v2.w, v2.w, s3);
      s0 = fmaf(v3.x, v3.x, s0); s1 = fmaf(v3.y, v3.y, s1); s2 = fmaf(v3.z, v3.z, s2); s3 = fmaf(v3.w, v3.w, s3);
    }
    for (; i < seg_end; i += 256) {
      const float4 vv = base[i];
      s0 = fmaf(vv.x, vv.x, s0); s1 = fmaf(vv.y, vv.y, s1); s2 = fmaf(vv.z, vv.z, s2); s3 = fmaf(vv.w, vv.w, s3);
    }
  }
  float sm = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                            const float* __restrict__ gscale) {
  const float gs = gscale ? gscale[0] : 1.0f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gs;
    const float mm = b1 * m[i] + (1.0f - b1) * gr;
    const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
    m[i] = mm;
    v[i] = vv;
    // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * mm / denom;
  }
}
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ x, long n, float* __restrict__ scratch) {
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long n4 = (((size_t)x & 15) == 0) ? n / 4 : 0;  // 16-byte loads when the buffer allows (the flat gradient buffer does)
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = x4[i];
    s0 = fmaf(v.x, v.x, s0); s1 = fmaf(v.y, v.y, s1); s2 = fmaf(v.z, v.z, s2); s3 = fmaf(v.w, v.w, s3);
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s0 += x[i] * x[i];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) scratch[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_stage2(const float* __restrict__ scratch, int nb, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += scratch[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}
// out[0] = sum of a[0 .. na) and b[0 .. nb) (block partials, then the extra terms), a fixed summation order; one block of 1024
__global__ __launch_bounds__(1024) void sumsq_stage2_two(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                         float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < na; i += 1024) s += a[i];
  int i = threadIdx.x;
  for (; i + 3072 < nb; i += 4096) {  // four loads in flight per lane
    const float v0 = b[i], v1 = b[i + 1024], v2 = b[i + 2048], v3 = b[i + 3072];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; i < nb; i += 1024) s += b[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) t += red[w];
    out[0] = t;
  }
}
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float inv_world, float* __restrict__ coef) {
  // gradients in the buffer are SUMS over `world` ranks: g = buf * inv_world.
  // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max = 1); coef[0] also folds 1/world.
  const float nrm = sqrtf(sumsq[0]) * inv_world;
  const float c = (max_norm > 0.f) ? fminf(1.0f, max_norm / (nrm + 1e-6f)) : 1.0f;
  coef[0] = c * inv_world;
  coef[1] = nrm;
}

// ---------------------------------------------------------------- probes (tests only)
__global__ void probe_tr16_kernel(const u16* __restrict__ in, const int* __restrict__ off, u16* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u16 lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + off[l]));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (u16)t[j];
}
__global__ void probe_mfma_kernel(int which, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c) {
  const int l = threadIdx.x;
  if (which == 0) {  // 16x16x32 bf16: a[16][32], b[32][16] -> c[16][16]
    bf16x8 af, bf;
    for (int i = 0; i < 8; i++) {
      af[i] = (__bf16)a[(l & 15) * 32 + (l >> 4) * 8 + i];
      bf[i] = (__bf16)b[((l >> 4) * 8 + i) * 16 + (l & 15)];
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) c[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
  } else {  // 32x32x16: a[32][16], b[16][32] -> c[32][32]
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    if (which == 1) {
      bf16x8 af, bf;
      for (int i = 0; i < 8; i++) {
        af[i] = (__bf16)a[(l & 31) * 16 + (l >> 5) * 8 + i];
        bf[i] = (__bf16)b[((l >> 5) * 8 + i) * 32 + (l & 31)];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    } else {
      f16x8 af, bf;
      for (int i = 0; i < 8; i++) {
        af[i] = (_Float16)a[(l & 31) * 16 + (l >> 5) * 8 + i];
        bf[i] = (_Float16)b[((l >> 5) * 8 + i) * 32 + (l & 31)];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) c[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
  }
}

inline int grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int vbx_pack_embed_input(const float* x, const float* cond, const uint8_t* cond_mask, void* out_f16, void* out_bf16,
                                    int B, int N, int D, void* stream) {
  VBX_REQUIRE(x && cond && out_f16 && D % 8 == 0, "vbx_pack_embed_input: bad args");
  const long rows = (long)B * N;
  hipLaunchKernelGGL(pack_embed_kernel, dim3(grid_for(rows * D / 4)), dim3(256), 0, ST, x, cond, cond_mask, (u16*)out_f16,
                     (u16*)out_bf16, rows, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_pack_embed_input_text(const float* x, const float* cond, const uint8_t* cond_mask, const uint8_t* drop_mask,
                                         const float* null_cond, const long* ids, int T, const float* table, int E, long null_id,
                                         void* out_f16, void* out_bf16, int B, int N, int D, void* stream) {
  VBX_REQUIRE(x && cond && ids && table && out_f16 && T > 0 && D % 8 == 0 && E > 0 && E % 8 == 0, "vbx_pack_embed_input_text: bad args");
  VBX_REQUIRE(!drop_mask || null_cond, "vbx_pack_embed_input_text: a drop mask needs null_cond");
  const long chunks = (long)B * N * (2 * D + E) / 8;
  hipLaunchKernelGGL(pack_embed_text_kernel, dim3(grid_for(chunks)), dim3(256), 0, ST, x, cond, cond_mask, drop_mask, null_cond, ids, T,
                     table, E, null_id, (u16*)out_f16, (u16*)out_bf16, (float*)nullptr, B, N, D);
  VBX_LAUNCH_CHECK();
  return 0;
}
// the same rows in fp32 (precise mode's to_embed operand)
extern "C" int vbx_embed_input_text_f32(const float* x, const float* cond, const uint8_t* cond_mask, const uint8_t* drop_mask,
                                        const float* null_cond, const long* ids, int T, const float* table, int E, long null_id,
                                        float* out_f32, int B, int N, int D, void* stream) {
  VBX_REQUIRE(x && cond && ids && table && out_f32 && T > 0 && D % 8 == 0 && E > 0 && E % 8 == 0, "vbx_embed_input_text_f32: bad args");
  VBX_REQUIRE(!drop_mask || null_cond, "vbx_embed_input_text_f32: a drop mask needs null_cond");
  const long chunks = (long)B * N * (2 * D + E) / 8;
  hipLaunchKernelGGL(pack_embed_text_kernel, dim3(grid_for(chunks)), dim3(256), 0, ST, x, cond, cond_mask, drop_mask, null_cond, ids, T,
                     table, E, null_id, (u16*)nullptr, (u16*)nullptr, out_f32, B, N, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_pack_phoneme_input(const long* ids, const float* table, int E, const float* cond, int S,
                                      const uint8_t* cond_mask, const uint8_t* drop_mask, const float* null_cond,
                                      void* out_f16, int B, int N, int D, void* stream) {
  VBX_REQUIRE(ids && table && cond && out_f16 && B > 0 && N > 0 && S > 0 && D % 8 == 0 && E > 0 && E % 8 == 0,
              "vbx_pack_phoneme_input: bad args");
  VBX_REQUIRE(!drop_mask || null_cond, "vbx_pack_phoneme_input: a drop mask needs null_cond");
  const long chunks = (long)B * N * (D + E) / 8;
  hipLaunchKernelGGL(pack_phoneme_kernel, dim3(grid_for(chunks)), dim3(256), 0, ST, ids, table, E, cond, S, cond_mask, drop_mask,
                     null_cond, (u16*)out_f16, B, N, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_rowdot(const float* x, const float* w, const float* bias, float* out, long rows, int D, void* stream) {
  VBX_REQUIRE(x && w && out && rows > 0 && D > 0 && D % 4 == 0, "vbx_rowdot: bad args");
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)cdiv(rows, 4L)), dim3(256), 0, ST, x, w, bias, out, rows, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_cond_emb_bwd(const void* demb_bf16, int ld, const long* ids, int T, const uint8_t* drop_mask, long null_id,
                                float* gtable, int B, int N, int E, void* stream) {
  VBX_REQUIRE(demb_bf16 && ids && gtable && T > 0 && E > 0 && ld >= E, "vbx_cond_emb_bwd: bad args");
  hipLaunchKernelGGL(cond_emb_bwd_kernel, dim3(grid_for((long)B * N * E)), dim3(256), 0, ST, (const u16*)demb_bf16, ld, ids, T,
                     drop_mask, null_id, gtable, B, N, E);
  VBX_LAUNCH_CHECK();
  return 0;
}

// The tap loops are fully unrolled over a compile-time kernel size (the weights live in registers): one instantiation per odd
// size up to 31 (the reference default, voicebox_pytorch.py:893; ConvPositionEmbed asserts an odd size, :211).
#define VBX_CONV_KS_SWITCH(ks, CALL)                                                                                      \
  switch (ks) {                                                                                                           \
    case 1: CALL(1); break;   case 3: CALL(3); break;   case 5: CALL(5); break;   case 7: CALL(7); break;                 \
    case 9: CALL(9); break;   case 11: CALL(11); break; case 13: CALL(13); break; case 15: CALL(15); break;               \
    case 17: CALL(17); break; case 19: CALL(19); break; case 21: CALL(21); break; case 23: CALL(23); break;               \
    case 25: CALL(25); break; case 27: CALL(27); break; case 29: CALL(29); break; default: CALL(31); break;               \
  }
static inline bool conv_ks_ok(int ks) { return ks >= 1 && ks <= 31 && (ks & 1); }

static int convpos_fwd_impl(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* reg, float* xs, int B,
                            int N, int R, int D, int ksize, bool libm, void* stream) {
  VBX_REQUIRE(e && w && bias && xs, "vbx_convpos_fwd: null pointer");
  VBX_REQUIRE(conv_ks_ok(ksize), "vbx_convpos_fwd: conv_pos_embed_kernel_size must be odd and <= 31 (got %d)", ksize);
  VBX_REQUIRE(R == 0 || reg, "vbx_convpos_fwd: register tokens missing");
  dim3 grid(cdiv(N, CT), cdiv(D, 64), B);
#define VBX_CALL(KS)                                                                                                           \
  if (libm)                                                                                                                    \
    hipLaunchKernelGGL((convpos_fwd_kernel<2, KS>), grid, dim3(256), (CT + KS - 1) * 64 * sizeof(float), ST, e, w, bias, mask, \
                       (const float*)nullptr, xs, N, R, D);                                                                    \
  else                                                                                                                         \
    hipLaunchKernelGGL((convpos_fwd_kernel<0, KS>), grid, dim3(256), (CT + KS - 1) * 64 * sizeof(float), ST, e, w, bias, mask, \
                       (const float*)nullptr, xs, N, R, D)
  VBX_CONV_KS_SWITCH(ksize, VBX_CALL)
#undef VBX_CALL
  VBX_LAUNCH_CHECK();
  if (R > 0) {
    hipLaunchKernelGGL(regs_fill_kernel, dim3(grid_for((long)B * R * D)), dim3(256), 0, ST, reg, xs, B, N + R, R, D);
    VBX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vbx_convpos_fwd(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* reg,
                               float* xs, int B, int N, int R, int D, int ksize, void* stream) {
  return convpos_fwd_impl(e, w, bias, mask, reg, xs, B, N, R, D, ksize, false, stream);
}
// the same with libm's erff in the GELU (the fast path's A-S form is 6e-7 = ten fp32 ulps off): precise mode
extern "C" int vbx_convpos_fwd_libm(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* reg,
                                    float* xs, int B, int N, int R, int D, int ksize, void* stream) {
  return convpos_fwd_impl(e, w, bias, mask, reg, xs, B, N, R, D, ksize, true, stream);
}

extern "C" int vbx_stack_input(const float* x, const float* reg, float* xs, int B, int N, int R, int D, void* stream) {
  VBX_REQUIRE(x && xs && B > 0 && N > 0 && R >= 0 && D % 4 == 0 && (R == 0 || reg), "vbx_stack_input: bad args");
  hipLaunchKernelGGL(stack_rows_kernel, dim3(grid_for((long)B * N * D / 4)), dim3(256), 0, ST, x, xs, B, N, R, D, 1);
  VBX_LAUNCH_CHECK();
  if (R > 0) {
    hipLaunchKernelGGL(regs_fill_kernel, dim3(grid_for((long)B * R * D)), dim3(256), 0, ST, reg, xs, B, N + R, R, D);
    VBX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vbx_unet_cat(const float* x, const float* skip, float scale, void* cat_f16, void* cat_bf16, long rows, int D, void* stream) {
  VBX_REQUIRE(x && skip && (cat_f16 || cat_bf16) && rows > 0 && D > 0 && D % 4 == 0, "vbx_unet_cat: bad args");
  hipLaunchKernelGGL(unet_cat_kernel, dim3(grid_for(rows * D / 2)), dim3(256), 0, ST, x, skip, scale, (u16*)cat_f16, (u16*)cat_bf16, rows, D);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_unet_split(const float* dcat, float scale, float* dx, void* dx_bf16, float* dskip, long rows, int D, void* stream) {
  VBX_REQUIRE(dcat && dx && dskip && rows > 0 && D > 0 && D % 4 == 0, "vbx_unet_split: bad args");
  hipLaunchKernelGGL(unet_split_kernel, dim3(grid_for(rows * D / 4)), dim3(256), 0, ST, dcat, scale, dx, (u16*)dx_bf16, dskip, rows, D);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_unet_addskip(float* dx, void* dx_bf16, const float* dskip, long n, void* stream) {
  VBX_REQUIRE(dx && dskip && n > 0 && n % 4 == 0, "vbx_unet_addskip: bad args");
  hipLaunchKernelGGL(unet_addskip_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ST, dx, (u16*)dx_bf16, dskip, n / 4);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_stack_input_bwd(const float* dxs, float* dx, float* dreg, int B, int N, int R, int D, void* stream) {
  VBX_REQUIRE(dxs && dx && B > 0 && N > 0 && R >= 0 && D % 4 == 0, "vbx_stack_input_bwd: bad args");
  hipLaunchKernelGGL(stack_rows_kernel, dim3(grid_for((long)B * N * D / 4)), dim3(256), 0, ST, dxs, dx, B, N, R, D, 0);
  VBX_LAUNCH_CHECK();
  if (R > 0 && dreg) {
    hipLaunchKernelGGL(dreg_kernel, dim3(cdiv((long)R * D, 256)), dim3(256), 0, ST, dxs, dreg, B, N + R, R, D);
    VBX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vbx_convpos_bwd_chunks(int B, int N) { return B * cdiv(N, CT); }

extern "C" int vbx_convpos_bwd(const float* e, const float* w, const float* bias, const uint8_t* mask, const float* dxs,
                               float* dpre_tmp, float* de, void* de_bf16, float* wpart, float* dreg, int B, int N, int R,
                               int D, int ksize, void* stream) {
  VBX_REQUIRE(e && w && bias && dxs && dpre_tmp && de && wpart, "vbx_convpos_bwd: null pointer");
  VBX_REQUIRE(conv_ks_ok(ksize), "vbx_convpos_bwd: conv_pos_embed_kernel_size must be odd and <= 31 (got %d)", ksize);
  dim3 grid(cdiv(N, CT), cdiv(D, 64), B);
  const int rows = CT + ksize - 1;
#define VBX_CALL(KS)                                                                                                            \
  hipLaunchKernelGGL((convpos_fwd_kernel<1, KS>), grid, dim3(256), rows * 64 * sizeof(float), ST, e, w, bias, mask, dxs, dpre_tmp, \
                     N, R, D)
  VBX_CONV_KS_SWITCH(ksize, VBX_CALL)
#undef VBX_CALL
  VBX_LAUNCH_CHECK();
  // the two tiles, or the [4][64][33] reduction buffer that reuses their space, whichever is larger
  size_t lds = (size_t)(2 * rows * 64) * sizeof(float);
  if (lds < (size_t)4 * 64 * 33 * sizeof(float)) lds = (size_t)4 * 64 * 33 * sizeof(float);
  static bool attr[32] = {};
#define VBX_CALL(KS)                                                                                                              \
  {                                                                                                                               \
    if (!attr[KS]) {                                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convpos_bwd_kernel<KS>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                                        \
      attr[KS] = true;                                                                                                            \
    }                                                                                                                             \
    hipLaunchKernelGGL(convpos_bwd_kernel<KS>, grid, dim3(256), lds, ST, e, w, mask, dxs, dpre_tmp, de, (u16*)de_bf16, wpart, N,   \
                       R, D);                                                                                                     \
  }
  VBX_CONV_KS_SWITCH(ksize, VBX_CALL)
#undef VBX_CALL
  VBX_LAUNCH_CHECK();
  if (R > 0 && dreg) {
    hipLaunchKernelGGL(dreg_kernel, dim3(cdiv((long)R * D, 256)), dim3(256), 0, ST, dxs, dreg, B, N + R, R, D);
    VBX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vbx_conv_wgrad_finalize(const float* wpart, int chunks, int D, int ksize, float* dw, float* db, void* stream) {
  VBX_REQUIRE(wpart && dw && db && ksize < 63, "vbx_conv_wgrad_finalize: bad args");
  hipLaunchKernelGGL(conv_wgrad_finalize_kernel, dim3(cdiv((long)D * 64, 256)), dim3(256), 0, ST, wpart, chunks, D, ksize, dw, db);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_time_embed_fwd(const float* times, const float* w_sin, const float* w1, const float* b1, float* four,
                                  float* pre, float* temb, int B, int D, int Th, void* stream) {
  VBX_REQUIRE(times && w_sin && w1 && b1 && four && pre && temb && D % 2 == 0, "vbx_time_embed_fwd: bad args");
  hipLaunchKernelGGL(time_four_kernel, dim3(cdiv((long)B * D, 256)), dim3(256), 0, ST, times, w_sin, four, B, D);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(time_linear_kernel, dim3(cdiv((long)B * Th, 4)), dim3(256), 0, ST, four, w1, b1, pre, temb, B, D, Th);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_time_embed_bwd_scratch_floats(int B, int D) { return (1 + TB_SLICES) * B * D; }

extern "C" int vbx_time_embed_bwd(const float* times, const float* w_sin, const float* w1, const float* four,
                                  const float* pre, const float* dtemb, float* dw_sin, float* dw1, float* db1,
                                  float* scratch, int B, int D, int Th, void* stream) {
  VBX_REQUIRE(times && w_sin && w1 && four && pre && dtemb && dw_sin && dw1 && db1 && scratch, "vbx_time_embed_bwd: null");
  hipLaunchKernelGGL(time_bwd_w1_kernel, dim3(cdiv((long)Th * D, 256)), dim3(256), 0, ST, four, pre, dtemb, dw1, db1, B, D, Th);
  VBX_LAUNCH_CHECK();
  // scratch: [0, B*D) d(four) ; [B*D, (1+TB_SLICES)*B*D) partials
  hipLaunchKernelGGL(time_bwd_four_kernel, dim3(cdiv(D, 64), B, TB_SLICES), dim3(64), 0, ST, w1, pre, dtemb, scratch + (long)B * D,
                     B, D, Th);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv((long)B * D, 64)), dim3(256), 0, ST, scratch + (long)B * D, (long)TB_SLICES,
                     (long)B * D, scratch, (long)B * D, 0);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(time_bwd_wsin_kernel, dim3(cdiv(D / 2, 256)), dim3(256), 0, ST, times, four, scratch, dw_sin, B, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_adaln_proj_fwd(const float* temb, const void* w_bf16, const float* bias, float* ada, int B, int Th, int J,
                                  int group, void* stream) {
  VBX_REQUIRE(temb && w_bf16 && bias && ada && Th % 8 == 0, "vbx_adaln_proj_fwd: bad args");
  if (group <= 0) group = J;
  VBX_REQUIRE(J % group == 0, "vbx_adaln_proj_fwd: J must be a multiple of group");
  if (B <= 16 && Th % 32 == 0) {  // the usual case: MFMA weight-streaming kernel
    hipLaunchKernelGGL(adaln_fwd_mfma_kernel, dim3(cdiv(J, 64)), dim3(256), 0, ST, temb, (const u16*)w_bf16, bias, ada, B, Th, J,
                       group);
    VBX_LAUNCH_CHECK();
    return 0;
  }
  int bc = B < 8 ? B : 8;
  while ((size_t)bc * Th * sizeof(float) > 128 * 1024 && bc > 1) bc >>= 1;
  const int lds = bc * Th * (int)sizeof(float);
  VBX_REQUIRE(lds <= 128 * 1024, "vbx_adaln_proj_fwd: time_hidden_dim too large (%d)", Th);
  static int attr_lds = 0;
  if (lds > attr_lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adaln_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_lds = lds;
  }
  hipLaunchKernelGGL(adaln_fwd_kernel, dim3(cdiv(J, 64)), dim3(256), lds, ST, temb, (const u16*)w_bf16, bias, ada, B, Th, J, bc, group);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_adaln_proj_bwd_scratch_floats(int B, int Th, int J) { return ADA_SLICES * B * Th; }

extern "C" int vbx_adaln_proj_bwd(const float* temb, const void* w_bf16, const float* dada, float* dw, float* dbias,
                                  float* dtemb, float* scratch, int B, int Th, int J, int accumulate_dtemb, void* stream) {
  VBX_REQUIRE(temb && w_bf16 && dada && dbias && dtemb && scratch && Th % 8 == 0, "vbx_adaln_proj_bwd: bad args");  // dw NULL: factor form (vbx_model.adaln_factors)
  static const int ver = getenv("VBX_ADALN_BWD") ? atoi(getenv("VBX_ADALN_BWD")) : 2;
  const dim3 grid(cdiv(Th / 4, 256), cdiv(J, ADA_WROWS) + ADA_SLICES);
  if (ver != 1 && J <= ADA_SLICES * ADA_PER_MAX)
    hipLaunchKernelGGL(adaln_bwd_kernel_v2, grid, dim3(256), 0, ST, temb, (const u16*)w_bf16, dada, dw, dbias, scratch, B, Th, J);
  else
    hipLaunchKernelGGL(adaln_bwd_kernel, grid, dim3(256), 0, ST, temb, (const u16*)w_bf16, dada, dw, dbias, scratch, B, Th, J);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv((long)B * Th, 64)), dim3(256), 0, ST, scratch, (long)ADA_SLICES,
                     (long)B * Th, dtemb, (long)B * Th, accumulate_dtemb);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" long vbx_adaln_dtemb_all_scratch_floats(int L, int B, int Th, int J) { return (long)L * cdiv(J, ADA_PER_MAX) * B * Th; }
extern "C" int vbx_adaln_dtemb_all(const void* w_f16, const float* dada, float* dtemb, float* scratch, int L, int B, int Th, int J,
                                   void* stream) {
  VBX_REQUIRE(w_f16 && dada && dtemb && scratch && L > 0 && B > 0 && J > 0 && Th % 8 == 0, "vbx_adaln_dtemb_all: bad args");
  const int spl = cdiv(J, ADA_PER_MAX);
  hipLaunchKernelGGL(adaln_dtemb_all_kernel, dim3(cdiv(Th / 8, 256), L * spl), dim3(256), 0, ST, (const u16*)w_f16, dada, scratch, B, Th, J,
                     spl);
  VBX_LAUNCH_CHECK();
  const long cols = (long)B * Th;
  hipLaunchKernelGGL(sum_rows_wide_kernel, dim3(cdiv(cols, 32)), dim3(1024), 0, ST, scratch, (long)L * spl, cols, dtemb, cols);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_sum_rows_f32(const float* in, long rows, long ld, float* out, long cols, int accumulate, void* stream) {
  VBX_REQUIRE(in && out && rows > 0 && cols > 0, "vbx_sum_rows_f32: bad args");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, ST, in, rows, ld, out, cols, accumulate);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_geglu_bwd(const void* h1_bf16, const void* dg_bf16, void* dh1_bf16, int M, int Fp, void* stream) {
  VBX_REQUIRE(h1_bf16 && dg_bf16 && dh1_bf16 && Fp % 64 == 0, "vbx_geglu_bwd: bad args (Fp must be a multiple of 64)");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for((long)M * Fp / 8)), dim3(256), 0, ST, (const u16*)h1_bf16,
                     (const u16*)dg_bf16, (u16*)dh1_bf16, (long)M, Fp);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_geglu_bwd_colsum_slabs(void) { return GB_SLABS; }
extern "C" int vbx_geglu_bwd_colsum(const void* h1_bf16, const void* dg_bf16, void* dh1_bf16, int M, int Fp, float* scratch,
                                    void* stream) {
  VBX_REQUIRE(h1_bf16 && dg_bf16 && dh1_bf16 && scratch && Fp % 64 == 0, "vbx_geglu_bwd_colsum: bad args (Fp must be a multiple of 64)");
  hipLaunchKernelGGL(geglu_bwd_colsum_kernel, dim3(cdiv(Fp / 8, 64), GB_SLABS), dim3(256), 0, ST, (const u16*)h1_bf16,
                     (const u16*)dg_bf16, (u16*)dh1_bf16, (long)M, Fp, scratch);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_multi_reduce(const vbx_mr_jobs* jobs, void* stream) {
  VBX_REQUIRE(jobs && jobs->n > 0 && jobs->n <= VBX_MR_MAX, "vbx_multi_reduce: bad job count");
  vbx_mr_jobs j = *jobs;
  int blocks = 0;
  for (int i = 0; i < j.n; i++) {
    VBX_REQUIRE(j.job[i].src && j.job[i].dst && j.job[i].rows > 0 && j.job[i].cols > 0 && j.job[i].batches > 0, "vbx_multi_reduce: bad job %d", i);
    j.job[i].block0 = blocks;
    blocks += j.job[i].batches * cdiv(j.job[i].cols, 64);
  }
  hipLaunchKernelGGL(multi_reduce_kernel, dim3(blocks), dim3(1024), 0, ST, j);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_layer_reduce(const vbx_skr_jobs* sjobs, const vbx_mr_jobs* mjobs, void* stream) {
  VBX_REQUIRE(sjobs && sjobs->n > 0 && sjobs->n <= VBX_SKR_MAX && mjobs && mjobs->n > 0 && mjobs->n <= VBX_MR_MAX,
              "vbx_layer_reduce: bad job counts");
  vbx_skr_jobs sj = *sjobs;
  vbx_mr_jobs mj = *mjobs;
  int sb = 0, mb = 0;
  for (int i = 0; i < sj.n; i++) {
    VBX_REQUIRE(sj.job[i].slabs && sj.job[i].dst && sj.job[i].splits >= 1 && sj.job[i].M > 0 && sj.job[i].N > 0 && sj.job[i].N % 4 == 0 &&
                    !sj.job[i].sq,
                "vbx_layer_reduce: bad split-K job %d (N must be a multiple of 4; sq partials are served by vbx_splitk_reduce_multi only)", i);
    sj.job[i].block0 = sb;
    sb += cdiv((long)sj.job[i].M * sj.job[i].N / 4, 1024);
  }
  for (int i = 0; i < mj.n; i++) {
    VBX_REQUIRE(mj.job[i].src && mj.job[i].dst && mj.job[i].rows > 0 && mj.job[i].cols > 0 && mj.job[i].batches > 0,
                "vbx_layer_reduce: bad reduction job %d", i);
    mj.job[i].block0 = mb;
    mb += mj.job[i].batches * cdiv(mj.job[i].cols, 64);
  }
  hipLaunchKernelGGL(layer_reduce_kernel, dim3(sb + mb), dim3(1024), 0, ST, sj, mj, sb);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_colsum_scratch_floats(int M, int C) { return CS_SLABS * C; }

extern "C" int vbx_colsum_bf16_partials(const void* in_bf16, int M, int C, int ld, float* scratch, void* stream) {
  VBX_REQUIRE(in_bf16 && scratch && C % 8 == 0 && ld % 8 == 0, "vbx_colsum_bf16_partials: bad args");
  hipLaunchKernelGGL(colsum_stage1<true>, dim3(cdiv(C, 64 * 8), CS_SLABS), dim3(256), 0, ST, in_bf16, (long)M, C, (long)ld, scratch);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_colsum_slabs(void) { return CS_SLABS; }
extern "C" int vbx_colsum_bf16(const void* in_bf16, int M, int C, int ld, float* out, int out_len, int rowmap, int F,
                               float* scratch, void* stream) {
  VBX_REQUIRE(in_bf16 && out && scratch, "vbx_colsum_bf16: null pointer");
  VBX_REQUIRE(C % 8 == 0 && ld % 8 == 0, "vbx_colsum_bf16: C and ld must be multiples of 8");
  hipLaunchKernelGGL(colsum_stage1<true>, dim3(cdiv(C, 64 * 8), CS_SLABS), dim3(256), 0, ST, in_bf16, (long)M, C, (long)ld, scratch);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(C, 256)), dim3(256), 0, ST, scratch, C, out, out_len, rowmap, F);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_colsum_f32(const float* in, int M, int C, int ld, float* out, float* scratch, void* stream) {
  VBX_REQUIRE(in && out && scratch, "vbx_colsum_f32: null pointer");
  VBX_REQUIRE(C % 4 == 0 && ld % 4 == 0, "vbx_colsum_f32: C and ld must be multiples of 4");
  hipLaunchKernelGGL(colsum_stage1<false>, dim3(cdiv(C, 64 * 4), CS_SLABS), dim3(256), 0, ST, (const void*)in, (long)M, C,
                     (long)ld, scratch);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(C, 256)), dim3(256), 0, ST, scratch, C, out, C, 0, 0);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_masked_mse_scratch_floats(int B) { return 2 * B + 2 * MSE_SPLITS * B; }

extern "C" int vbx_masked_mse_fwd(const float* pred, const float* target, const uint8_t* loss_mask, float* per_b, float* loss,
                                  int B, int N, int D, void* stream) {
  VBX_REQUIRE(pred && target && loss_mask && per_b && loss && D % 4 == 0, "vbx_masked_mse_fwd: bad args");
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(MSE_SPLITS, B), dim3(256), 0, ST, pred, target, loss_mask, per_b, B, N, D);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(mse_mean_kernel, dim3(1), dim3(64), 0, ST, per_b, loss, B);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_masked_mse_bwd(const float* pred, const float* target, const uint8_t* loss_mask, const float* per_b,
                                  const float* gscale, float* dpred, void* dpred_bf16, int B, int N, int D, void* stream) {
  VBX_REQUIRE(pred && target && loss_mask && per_b && (dpred || dpred_bf16) && D % 4 == 0, "vbx_masked_mse_bwd: bad args");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for((long)B * N * D / 4)), dim3(256), 0, ST, pred, target, loss_mask, per_b,
                     gscale, dpred, (u16*)dpred_bf16, B, N, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_cfm_inputs(const float* x1, const float* x0, const float* times, float sigma, float* w, float* flow, int B,
                              long per_batch, void* stream) {
  VBX_REQUIRE(x1 && x0 && times && w && flow, "vbx_cfm_inputs: null pointer");
  hipLaunchKernelGGL(cfm_inputs_kernel, dim3(grid_for((long)B * per_batch)), dim3(256), 0, ST, x1, x0, times, sigma, w, flow,
                     B, per_batch);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_axpy_dev(const float* y, const float* f, const float* coef, int idx, float* out, long n, void* stream) {
  VBX_REQUIRE(y && f && coef && out && n % 4 == 0, "vbx_axpy_dev: bad args");
  hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ST, y, f, coef, idx, out, n / 4);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_ode_set_time(float* times, int B, const float* table, const int* counter, int slot, void* stream) {
  VBX_REQUIRE(times && table && counter && (slot == 0 || slot == 1), "vbx_ode_set_time: bad args");
  hipLaunchKernelGGL(ode_set_time_kernel, dim3(cdiv(B, 64)), dim3(64), 0, ST, times, B, table, counter, slot);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_ada_select(float* ada, int L, int B, int G, const float* table, const int* counter, int slot, void* stream) {
  VBX_REQUIRE(ada && table && counter && L > 0 && B > 0 && G > 0 && G % 4 == 0 && (slot == 0 || slot == 1), "vbx_ada_select: bad args");
  hipLaunchKernelGGL(ada_select_kernel, dim3(grid_for((long)L * G / 4, 256)), dim3(256), 0, ST, ada, L, B, G, table, counter, slot);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_axpy_ctr(const float* y, const float* f, const float* table, const int* counter, int slot, float* out,
                            long n, void* stream) {
  VBX_REQUIRE(y && f && table && counter && out && n % 4 == 0, "vbx_axpy_ctr: bad args");
  hipLaunchKernelGGL(axpy_ctr_kernel, dim3(grid_for(n / 4)), dim3(256), 0, ST, y, f, table, counter, slot, out, n / 4);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_counter_add(int* counter, int inc, void* stream) {
  VBX_REQUIRE(counter, "vbx_counter_add: null");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, ST, counter, inc);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_stream_delay(float us, void* stream) {
  VBX_REQUIRE(us >= 0.f && us <= 1e4f, "vbx_stream_delay: 0 .. 10000 us");
  if (us > 0.f) {
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, ST, (unsigned long long)(us * 100.0f));
    VBX_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vbx_pack_weight(const float* src, int src_rows, int src_cols, void* dst_bf16, void* dst_f16, int dst_rows,
                               int dst_cols, int rowmap, int F, void* stream) {
  VBX_REQUIRE(src && (dst_bf16 || dst_f16), "vbx_pack_weight: null pointer");
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for((long)dst_rows * dst_cols)), dim3(256), 0, ST, src, src_rows, src_cols,
                     (u16*)dst_bf16, (u16*)dst_f16, dst_rows, dst_cols, rowmap, F);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_pack_bias(const float* src, int n, float* dst, int dst_n, int rowmap, int F, void* stream) {
  VBX_REQUIRE(src && dst, "vbx_pack_bias: null pointer");
  hipLaunchKernelGGL(pack_bias_kernel, dim3(cdiv(dst_n, 256)), dim3(256), 0, ST, src, n, dst, dst_n, rowmap, F);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_adam_step_packed(float* p, const float* g, float* m, float* v, const vbx_adam_seg* segs_dev, int nsegs,
                                    long total_blocks, float lr, float beta1, float beta2, float eps, int step,
                                    const float* gscale, void* stream) {
  VBX_REQUIRE(p && g && m && v && segs_dev && nsegs > 0 && total_blocks > 0 && step >= 1, "vbx_adam_step_packed: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_packed_kernel, dim3((unsigned)total_blocks), dim3(256), 0, ST, p, g, m, v, segs_dev, nsegs, lr / bc1, beta1,
                     beta2, eps, sqrtf(bc2), gscale);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, int step, const float* gscale, void* stream) {
  VBX_REQUIRE(p && g && m && v && n > 0 && step >= 1, "vbx_adam_step: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, ST, p, g, m, v, n, lr, beta1, beta2, eps, bc1,
                     sqrtf(bc2), gscale);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_sumsq(const float* x, long n, float* out, float* scratch, void* stream) {
  VBX_REQUIRE(x && out && scratch && n > 0, "vbx_sumsq: bad args");
  const int nb = grid_for(n, 1024);
  hipLaunchKernelGGL(sumsq_stage1, dim3(nb), dim3(256), 0, ST, x, n, scratch);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, ST, scratch, nb, out);
  VBX_LAUNCH_CHECK();
  return 0;
}
// sum of squares over n <= 64 ranges [lo_k, hi_k) of x (floats; multiples of 4) plus n_extra values already sitting in
// scratch[1024 .. 1024 + n_extra) (the factor-form terms of vbx_sumsq_adaln_factors, the slab-reduce partials of vbx_skr_job.sq)
// -> out[0].  scratch >= 1024 + n_extra floats.
extern "C" int vbx_sumsq_ranges(const float* x, const long* ranges /* host [2 n] */, int n, int n_extra, float* out, float* scratch,
                                void* stream) {
  VBX_REQUIRE(x && ranges && out && scratch && n > 0 && n <= 64 && n_extra >= 0 && ((size_t)x & 15) == 0, "vbx_sumsq_ranges: bad args");
  SumsqRanges rg;
  rg.n = n;
  rg.pre[0] = 0;
  for (int k = 0; k < n; k++) {
    const long lo = ranges[2 * k], hi = ranges[2 * k + 1];
    VBX_REQUIRE(lo >= 0 && hi >= lo && lo % 4 == 0 && hi % 4 == 0, "vbx_sumsq_ranges: range %d [%ld, %ld) must be 4-float aligned", k, lo, hi);
    rg.lo[k] = lo;
    rg.pre[k + 1] = rg.pre[k] + (hi - lo) / 4;
  }
  const int nb = grid_for(rg.pre[n] * 4, 1024);
  hipLaunchKernelGGL(sumsq_ranges_stage1, dim3(nb), dim3(256), 0, ST, x, rg, scratch);
  VBX_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2_two, dim3(1), dim3(1024), 0, ST, scratch, nb, scratch + 1024, n_extra, out);
  VBX_LAUNCH_CHECK();
  return 0;
}
// factor-form adaLN weight gradients (see adam_adaln_factor_kernel): Adam on the L weight blocks, and their sums of squares
extern "C" int vbx_adam_adaln_factors(float* p, float* m, float* v, const long* w_off /* host [L] */, void* const* dst_f16 /* host [L] */,
                                      const float* dada, const float* temb, int L, int B, int J4, int Th, float lr, float beta1,
                                      float beta2, float eps, int step, const float* gscale, void* stream) {
  VBX_REQUIRE(p && m && v && w_off && dada && temb && L > 0 && L <= 32 && B > 0 && J4 > 0 && Th > 0 && Th % 4 == 0 && step >= 1,
              "vbx_adam_adaln_factors: bad args");
  AdaFactorArgs a{};
  for (int l = 0; l < L; l++) {
    VBX_REQUIRE(w_off[l] % 4 == 0, "vbx_adam_adaln_factors: weight offsets must be 16-byte aligned");
    a.w_off[l] = w_off[l];
    a.dst_f16[l] = dst_f16 ? (u16*)dst_f16[l] : nullptr;
  }
  a.dada = dada; a.temb = temb; a.L = L; a.B = B; a.J4 = J4; a.Th = Th;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_adaln_factor_kernel, dim3(cdiv(Th, 1024), cdiv(J4, ADAF_ROWS), L), dim3(256), 0, ST, p, m, v, a, lr / bc1,
                     beta1, beta2, eps, sqrtf(bc2), gscale);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_adaln_expand_dw(const float* temb, const float* dada, float* dw, int B, int Th, int J4, int reserved, void* stream) {
  (void)reserved;
  VBX_REQUIRE(temb && dada && dw && B > 0 && Th > 0 && Th % 4 == 0 && J4 > 0, "vbx_adaln_expand_dw: bad args");
  hipLaunchKernelGGL(adaln_expand_dw_kernel, dim3(cdiv(Th / 4, 256), cdiv(J4, ADA_WROWS)), dim3(256), 0, ST, temb, dada, dw, B, Th, J4);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_sumsq_adaln_factors(const float* dada, const float* temb, int L, int B, int J4, int Th, float* out /* [L * B * B] */,
                                       void* stream) {
  VBX_REQUIRE(dada && temb && out && L > 0 && B > 0 && B <= 64, "vbx_sumsq_adaln_factors: bad args");
  hipLaunchKernelGGL(adaln_factor_sumsq_kernel, dim3(B * B, L), dim3(256), 0, ST, dada, temb, out, B, J4, Th);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_clip_coef(const float* sumsq, float max_norm, float inv_world, float* coef, void* stream) {
  VBX_REQUIRE(sumsq && coef, "vbx_clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, ST, sumsq, max_norm, inv_world, coef);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_probe_tr16(const void* in_u16_4096, const int* lane_elem_off, void* out_u16_256, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, ST, (const u16*)in_u16_4096, lane_elem_off, (u16*)out_u16_256);
  VBX_LAUNCH_CHECK();
  return 0;
}
extern "C" int vbx_probe_mfma(int which, const float* a, const float* b, float* c, void* stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, ST, which, a, b, c);
  VBX_LAUNCH_CHECK();
  return 0;
}

// ============================================================================ dropout (attend.py:131, voicebox_pytorch.py:346)
// Training-time dropout of the attention probabilities and of the GEGLU output.  The mask is a pure function of
// (seed, stream, element index) through Philox4x32-10 (common.hpp): nothing is drawn from a stateful generator on the device.
namespace {

// Attention keep bits of one layer, in BOTH orientations the attention kernels read them in:
//   R[bh][q][W2]    bit (key % 32) of word key / 32  -- forward and the dq body: a lane owns one query, its registers 32 keys
//   C[bh][key][W2]  bit (q % 32)   of word q / 32    -- the dk/dv body: a lane owns one key, its registers 32 queries
// W2 = 2 * ceil(Np / 64) words per row (even, so a 64-wide tile's two words are one aligned 8-byte load); bits past Np are 0.
// One workgroup = 64 queries of one head; wave w walks the key words w, w + 4, ...; a lane draws the 32 keys of its query
// (4 Philox calls) and the wave transposes the 64 x 32 bit block with 32 ballots.
__global__ __launch_bounds__(256) void attn_dropout_bits_kernel(unsigned* __restrict__ R, unsigned* __restrict__ C, int Np, int W2,
                                                                unsigned k0, unsigned k1, unsigned stream_id, unsigned thr16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane, bh = blockIdx.y;
  for (int kw = wave; kw < W2; kw += 4) {
    unsigned word = 0;
    const int valid = Np - kw * 32;
    if (q < Np && valid > 0) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        word |= philox_keep8(philox4x32_10((unsigned)(kw * 4 + i), (unsigned)q, (unsigned)bh, stream_id, k0, k1), thr16) << (8 * i);
      if (valid < 32) word &= (1u << valid) - 1u;
    }
    if (q < Np) R[((long)bh * Np + q) * W2 + kw] = word;
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const unsigned long long m = __ballot((word >> j) & 1u);
      if (lane == j) mine = m;
    }
    const int key = kw * 32 + lane;
    if (lane < 32 && key < Np)
      *reinterpret_cast<uint2*>(C + ((long)bh * Np + key) * W2 + blockIdx.x * 2) = make_uint2((unsigned)mine, (unsigned)(mine >> 32));
  }
}

// In-place dropout of a [rows, cols] 16-bit matrix (row stride ld) held as an fp16 copy and / or a bf16 copy: 8 elements
// (one Philox call, one 16-byte access per copy) per thread.  cols, ld multiples of 8.
__global__ __launch_bounds__(256) void dropout_rows_kernel(u16* __restrict__ xh, u16* __restrict__ xb, float* __restrict__ x32, long rows,
                                                           int cols, int ld, unsigned k0, unsigned k1, unsigned stream_id, unsigned thr16, float rkeep) {
  const int c8 = cols >> 3;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * c8) return;
  const long row = idx / c8;
  const int cb = (int)(idx - row * c8);
  const unsigned keep = philox_keep8(philox4x32_10((unsigned)cb, (unsigned)row, (unsigned)(row >> 32), stream_id, k0, k1), thr16);
  const long off = row * ld + cb * 8;
  if (xh) {
    uint4 v = *reinterpret_cast<const uint4*>(xh + off);
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float lo = ((keep >> (2 * i)) & 1u) ? f16_to_f32((u16)(w[i] & 0xFFFFu)) * rkeep : 0.f;
      const float hi = ((keep >> (2 * i + 1)) & 1u) ? f16_to_f32((u16)(w[i] >> 16)) * rkeep : 0.f;
      w[i] = pack_f16x2_sat(lo, hi);
    }
    *reinterpret_cast<uint4*>(xh + off) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (xb) {
    uint4 v = *reinterpret_cast<const uint4*>(xb + off);
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float lo = ((keep >> (2 * i)) & 1u) ? bf16_to_f32((u16)(w[i] & 0xFFFFu)) * rkeep : 0.f;
      const float hi = ((keep >> (2 * i + 1)) & 1u) ? bf16_to_f32((u16)(w[i] >> 16)) * rkeep : 0.f;
      w[i] = pack_bf16x2(lo, hi);
    }
    *reinterpret_cast<uint4*>(xb + off) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (x32) {  // precise mode: the unrounded copy takes the same mask
#pragma unroll
    for (int i = 0; i < 8; i++) x32[off + i] = ((keep >> i) & 1u) ? x32[off + i] * rkeep : 0.f;
  }
}

}  // namespace

extern "C" int vbx_dropout_bits_words(int Np) { return Np > 0 ? 2 * cdiv(Np, 64) : 0; }

extern "C" int vbx_attn_dropout_bits(void* bits_rm, void* bits_cm, int BH, int Np, unsigned long long seed, unsigned stream_id,
                                     float p, void* stream) {
  VBX_REQUIRE(bits_rm && bits_cm && BH > 0 && BH <= 65535 && Np > 0, "vbx_attn_dropout_bits: bad args");
  VBX_REQUIRE(p > 0.f && p < 1.f, "vbx_attn_dropout_bits: p must be in (0, 1)");
  hipLaunchKernelGGL(attn_dropout_bits_kernel, dim3(cdiv(Np, 64), BH), dim3(256), 0, ST, (unsigned*)bits_rm, (unsigned*)bits_cm, Np,
                     vbx_dropout_bits_words(Np), (unsigned)seed, (unsigned)(seed >> 32), stream_id, dropout_thr16(p));
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" float vbx_dropout_keep_scale(float p) { return 65536.0f / (float)dropout_thr16(p); }

extern "C" int vbx_dropout_rows(void* x_f16, void* x_bf16, long rows, int cols, int ld, unsigned long long seed, unsigned stream_id,
                                float p, void* stream) {
  VBX_REQUIRE((x_f16 || x_bf16) && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "vbx_dropout_rows: bad args (cols, ld multiples of 8)");
  VBX_REQUIRE(p > 0.f && p < 1.f, "vbx_dropout_rows: p must be in (0, 1)");
  const long n = rows * (cols / 8);
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (u16*)x_f16, (u16*)x_bf16, (float*)nullptr, rows, cols, ld,
                     (unsigned)seed, (unsigned)(seed >> 32), stream_id, dropout_thr16(p), vbx_dropout_keep_scale(p));
  VBX_LAUNCH_CHECK();
  return 0;
}
// the same mask on an fp32 matrix (precise mode keeps the GEGLU output unrounded)
extern "C" int vbx_dropout_rows_f32(float* x, long rows, int cols, int ld, unsigned long long seed, unsigned stream_id, float p,
                                    void* stream) {
  VBX_REQUIRE(x && rows > 0 && cols > 0 && cols % 8 == 0 && ld % 8 == 0 && ld >= cols, "vbx_dropout_rows_f32: bad args (cols, ld multiples of 8)");
  VBX_REQUIRE(p > 0.f && p < 1.f, "vbx_dropout_rows_f32: p must be in (0, 1)");
  const long n = rows * (cols / 8);
  hipLaunchKernelGGL(dropout_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (u16*)nullptr, (u16*)nullptr, x, rows, cols, ld,
                     (unsigned)seed, (unsigned)(seed >> 32), stream_id, dropout_thr16(p), vbx_dropout_keep_scale(p));
  VBX_LAUNCH_CHECK();
  return 0;
}
