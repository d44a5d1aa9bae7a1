// GateLoop layer (gateloop_transformer.SimpleGateLoopLayer(dim, post_ln=True); call sites voicebox_pytorch.py:399,465-466):
//   x^ = RMSNorm(x);  q|kv|a = x^ W^T (one [3D,D] Linear, no bias);  a = sigmoid(a)
//   h_t = a_t h_{t-1} + kv_t  (per (batch, channel), h_{-1} = 0);  s_t = q_t h_t;  out = LayerNorm(s)
// The norm and the projection reuse the RMSNorm / GEMM kernels; this file holds the scan and the post-LayerNorm.
//
// HBM-bound work: the scan touches 3 (forward) / 6 (backward) fp32 streams of [B*Np, D].  A sequential scan over Np
// per channel would run B*D = 4096 threads at memory latency, so the frame axis is cut into GL_CHUNKS chunks that run
// as threads of the same block: pass 1 reduces each chunk to its affine map h -> P*h + H (two numbers), a <= 31-step
// fold over LDS gives each chunk its carry-in, pass 2 replays the chunk from the carry (the re-read hits L2).
// A block owns 32 channels of one batch element -> every wave-instruction reads two 128-byte segments.
#include "common.hpp"

namespace {

constexpr int GL_CH = 32;      // channels per block
constexpr int GL_CHUNKS = 32;  // frame chunks per block (threads = GL_CH * GL_CHUNKS = 1024)

VBX_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// qkva [B, Np, 3D] fp32 (q | kv | a-logit) -> s [B, Np, D] (and h [B, Np, D] when hst != NULL: kept for backward)
__global__ __launch_bounds__(GL_CH* GL_CHUNKS) void gl_scan_fwd_kernel(const float* __restrict__ qkva, float* __restrict__ s,
                                                                        float* __restrict__ hst, int Np, int D) {
  __shared__ float sP[GL_CHUNKS][GL_CH], sH[GL_CHUNKS][GL_CH];
  const int c = threadIdx.x & (GL_CH - 1), ch = threadIdx.x / GL_CH;
  const int d = blockIdx.x * GL_CH + c, b = blockIdx.y;
  const int T = (Np + GL_CHUNKS - 1) / GL_CHUNKS;
  const int t0 = ch * T, t1 = min(Np, t0 + T);
  const bool live = d < D;
  const long D3 = 3L * D;
  const float* base = qkva + (long)b * Np * D3 + d;
  float P = 1.f, H = 0.f;
  if (live)
    for (int t = t0; t < t1; t++) {
      const float a = sigmoidf_(base[t * D3 + 2 * D]);
      H = fmaf(a, H, base[t * D3 + D]);
      P *= a;
    }
  sP[ch][c] = P;
  sH[ch][c] = H;
  __syncthreads();
  float h = 0.f;
  for (int j = 0; j < ch; j++) h = fmaf(sP[j][c], h, sH[j][c]);
  if (!live) return;
  float* so = s + (long)b * Np * D + d;
  float* ho = hst ? hst + (long)b * Np * D + d : nullptr;
  for (int t = t0; t < t1; t++) {
    const float a = sigmoidf_(base[t * D3 + 2 * D]);
    h = fmaf(a, h, base[t * D3 + D]);
    so[(long)t * D] = base[t * D3] * h;
    if (ho) ho[(long)t * D] = h;
  }
}

// Backward of the scan.  With u_t = ds_t q_t and g_t = dL/dh_t:  g_t = u_t + a_{t+1} g_{t+1}  (g_Np = 0);
//   dq_t = ds_t h_t;  dkv_t = g_t;  da_t = g_t h_{t-1};  d(a-logit)_t = da_t a_t (1 - a_t).
// The reverse recurrence is chunked exactly like the forward one (coefficients a_{t+1}).  Output bf16 [B, Np, 3D]:
// the operand of the dgrad / wgrad GEMMs of the projection.
__global__ __launch_bounds__(GL_CH* GL_CHUNKS) void gl_scan_bwd_kernel(const float* __restrict__ qkva, const float* __restrict__ hst,
                                                                        const float* __restrict__ ds, u16* __restrict__ dqkva,
                                                                        int Np, int D) {
  __shared__ float sP[GL_CHUNKS][GL_CH], sG[GL_CHUNKS][GL_CH];
  const int c = threadIdx.x & (GL_CH - 1), ch = threadIdx.x / GL_CH;
  const int d = blockIdx.x * GL_CH + c, b = blockIdx.y;
  const int T = (Np + GL_CHUNKS - 1) / GL_CHUNKS;
  const int t0 = ch * T, t1 = min(Np, t0 + T);
  const bool live = d < D;
  const long D3 = 3L * D;
  const float* base = qkva + (long)b * Np * D3 + d;
  const float* dsb = ds + (long)b * Np * D + d;
  float P = 1.f, G = 0.f;
  if (live && t0 < t1) {
    float an = (t1 < Np) ? sigmoidf_(base[t1 * D3 + 2 * D]) : 0.f;  // a_{t+1} of the chunk's last frame
    for (int t = t1 - 1; t >= t0; t--) {
      G = fmaf(an, G, dsb[(long)t * D] * base[t * D3]);
      P *= an;
      an = sigmoidf_(base[t * D3 + 2 * D]);
    }
  }
  sP[ch][c] = P;
  sG[ch][c] = G;
  __syncthreads();
  float g = 0.f;  // g at the first frame of chunk ch+1 = carry-in of this chunk
  for (int j = GL_CHUNKS - 1; j > ch; j--) g = fmaf(sP[j][c], g, sG[j][c]);
  if (!live || t0 >= t1) return;
  const float* hb = hst + (long)b * Np * D + d;
  u16* ob = dqkva + (long)b * Np * D3 + d;
  float an = (t1 < Np) ? sigmoidf_(base[t1 * D3 + 2 * D]) : 0.f;
  float h = hb[(long)(t1 - 1) * D];
  for (int t = t1 - 1; t >= t0; t--) {
    const float dst = dsb[(long)t * D];
    const float a = sigmoidf_(base[t * D3 + 2 * D]);
    const float hprev = t > 0 ? hb[(long)(t - 1) * D] : 0.f;
    g = fmaf(an, g, dst * base[t * D3]);
    ob[t * D3] = f32_to_bf16(dst * h);
    ob[t * D3 + D] = f32_to_bf16(g);
    ob[t * D3 + 2 * D] = f32_to_bf16(g * hprev * a * (1.f - a));
    an = a;
    h = hprev;
  }
}

// ------------------------------------------------------------------ post LayerNorm (nn.LayerNorm(D), eps 1e-5) + residual
template <int NC>  // float4 chunks per lane actually needed (2: D <= 512, 4: D <= 1024, 8: D <= 2048), see norm.hip
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ resid,
                                                             float* __restrict__ y, long rows, int D, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const float invD = 1.0f / (float)D;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  const float4* b4 = reinterpret_cast<const float4*>(bias);
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float4* sr = reinterpret_cast<const float4*>(s + r * D);
    float4 v[NC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        v[i] = sr[c];
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mean = wave_sum(sum) * invD;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        var += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(var) * invD + eps);
    const float4* rr = resid ? reinterpret_cast<const float4*>(resid + r * D) : nullptr;
    float4* yr = reinterpret_cast<float4*>(y + r * D);
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        const float4 g = w4[c], bb = b4[c];
        float4 o = make_float4(v[i].x * rstd * g.x + bb.x, v[i].y * rstd * g.y + bb.y, v[i].z * rstd * g.z + bb.z,
                               v[i].w * rstd * g.w + bb.w);
        if (rr) {
          const float4 a = rr[c];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        yr[c] = o;
      }
    }
  }
}

// xh = (s - mean) rstd;  dw += dy xh;  db += dy;  dxh = dy w;  ds = rstd (dxh - mean(dxh) - xh mean(dxh xh))
// grid (chunks of 16 rows, B); partial records part[b][chunk][2][D] (dw | db), reduced by vbx_reduce_norm_partials.
constexpr int LN_WAVES = 8;
template <int NC>
__global__ __launch_bounds__(64 * LN_WAVES) void layernorm_bwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                                       const float* __restrict__ dy, float* __restrict__ ds,
                                                                       float* __restrict__ part, int Np, int D, float eps) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [LN_WAVES][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const float invD = 1.0f / (float)D;
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float4 aw[NC], ab[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) { aw[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); }
  for (int k = 0; k < 16 / LN_WAVES; k++) {
    const int j = blockIdx.x * 16 + wave + LN_WAVES * k;
    if (j >= Np) break;
    const long r = (long)blockIdx.y * Np + j;
    const float4* sr = reinterpret_cast<const float4*>(s + r * D);
    const float4* dr = reinterpret_cast<const float4*>(dy + r * D);
    float4 v[NC], g[NC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        v[i] = sr[c];
        g[i] = dr[c];
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mean = wave_sum(sum) * invD;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        var += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(var) * invD + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4) {
        const float4 ww = w4[c];
        v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xh
        aw[i].x += g[i].x * v[i].x; aw[i].y += g[i].y * v[i].y; aw[i].z += g[i].z * v[i].z; aw[i].w += g[i].w * v[i].w;
        ab[i].x += g[i].x; ab[i].y += g[i].y; ab[i].z += g[i].z; ab[i].w += g[i].w;
        g[i].x *= ww.x; g[i].y *= ww.y; g[i].z *= ww.z; g[i].w *= ww.w;  // dxh
        m1 += g[i].x + g[i].y + g[i].z + g[i].w;
        m2 += g[i].x * v[i].x + g[i].y * v[i].y + g[i].z * v[i].z + g[i].w * v[i].w;
      }
    }
    m1 = wave_sum(m1) * invD;
    m2 = wave_sum(m2) * invD;
    float4* dsr = reinterpret_cast<float4*>(ds + r * D);
#pragma unroll
    for (int i = 0; i < NC; i++) {
      const int c = lane + 64 * i;
      if (c < D4)
        dsr[c] = make_float4(rstd * (g[i].x - m1 - v[i].x * m2), rstd * (g[i].y - m1 - v[i].y * m2),
                             rstd * (g[i].z - m1 - v[i].z * m2), rstd * (g[i].w - m1 - v[i].w * m2));
    }
  }
  float4* r4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int i = 0; i < NC; i++) {
    const int c = lane + 64 * i;
    if (c < D4) {
      r4[(wave * 2 + 0) * D4 + c] = aw[i];
      r4[(wave * 2 + 1) * D4 + c] = ab[i];
    }
  }
  __syncthreads();
  float4* p4 = reinterpret_cast<float4*>(part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * D);
  for (int idx = threadIdx.x; idx < 2 * D4; idx += 64 * LN_WAVES) {
    const int which = idx / D4, c = idx - which * D4;
    float4 t = r4[which * D4 + c];
#pragma unroll
    for (int wv = 1; wv < LN_WAVES; wv++) {
      const float4 u = r4[(wv * 2 + which) * D4 + c];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    p4[which * D4 + c] = t;
  }
}

}  // namespace

extern "C" int vbx_gateloop_scan_fwd(const float* qkva, float* s, float* hstate, int B, int Np, int D, void* stream) {
  VBX_REQUIRE(qkva && s && B > 0 && Np > 0 && D > 0, "vbx_gateloop_scan_fwd: bad args");
  hipLaunchKernelGGL(gl_scan_fwd_kernel, dim3(cdiv(D, GL_CH), B), dim3(GL_CH * GL_CHUNKS), 0, (hipStream_t)stream, qkva, s, hstate,
                     Np, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_gateloop_scan_bwd(const float* qkva, const float* hstate, const float* ds, void* dqkva_bf16, int B, int Np, int D,
                                     void* stream) {
  VBX_REQUIRE(qkva && hstate && ds && dqkva_bf16 && B > 0 && Np > 0 && D > 0, "vbx_gateloop_scan_bwd: bad args");
  hipLaunchKernelGGL(gl_scan_bwd_kernel, dim3(cdiv(D, GL_CH), B), dim3(GL_CH * GL_CHUNKS), 0, (hipStream_t)stream, qkva, hstate, ds,
                     (u16*)dqkva_bf16, Np, D);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_layernorm_fwd(const float* s, const float* w, const float* bias, const float* resid, float* y, long rows, int D,
                                 float eps, void* stream) {
  VBX_REQUIRE(s && w && bias && y && rows > 0 && D > 0 && D % 4 == 0 && D <= 2048, "vbx_layernorm_fwd: bad args (D %% 4, D <= 2048)");
  const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
#define VBX_LNF(NC_) hipLaunchKernelGGL((layernorm_fwd_kernel<NC_>), dim3(grid), dim3(256), 0, (hipStream_t)stream, s, w, bias, resid, y, rows, D, eps)
  if (D <= 512) VBX_LNF(2);
  else if (D <= 1024) VBX_LNF(4);
  else VBX_LNF(8);
#undef VBX_LNF
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_layernorm_bwd(const float* s, const float* w, const float* dy, float* ds, float* part /* [B][ceil(Np/16)][2][D] */,
                                 int B, int Np, int D, float eps, void* stream) {
  VBX_REQUIRE(s && w && dy && ds && part && B > 0 && Np > 0 && D > 0 && D % 4 == 0 && D <= 2048, "vbx_layernorm_bwd: bad args");
  const size_t lds = (size_t)LN_WAVES * 2 * D * sizeof(float);
#define VBX_LNB(NC_)                                                                                                          \
  do {                                                                                                                       \
    static bool attr = false;                                                                                                \
    if (lds > 48 * 1024 && !attr) {                                                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(layernorm_bwd_kernel<NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                                                 \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((layernorm_bwd_kernel<NC_>), dim3(cdiv(Np, 16), B), dim3(64 * LN_WAVES), lds, (hipStream_t)stream, s, w, dy, \
                       ds, part, Np, D, eps);                                                                                \
  } while (0)
  if (D <= 512) VBX_LNB(2);
  else if (D <= 1024) VBX_LNB(4);
  else VBX_LNB(8);
#undef VBX_LNB
  VBX_LAUNCH_CHECK();
  return 0;
}
