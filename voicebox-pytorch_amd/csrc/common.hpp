// Shared device/host helpers for libvbx_hip.so (gfx950 only -- no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/vbx.h"

typedef unsigned short u16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VBX_DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---- host side error plumbing -------------------------------------------------------------
void vbx_set_error(const char* fmt, ...);
#define VBX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      vbx_set_error(__VA_ARGS__);         \
      return VBX_EINVAL;                  \
    }                                     \
  } while (0)
#define VBX_LAUNCH_CHECK()                                                  \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vbx_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return (int)e__;                                                      \
    }                                                                       \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- bf16 / fp16 scalar helpers -----------------------------------------------------------
VBX_DEV float bf16_to_f32(u16 v) { return __uint_as_float(((unsigned)v) << 16); }
VBX_DEV u16 f32_to_bf16(float f) {  // round-to-nearest-even (NaN preserved)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(u16, b);
}
VBX_DEV unsigned pack_bf16x2(float lo, float hi) { return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16); }
VBX_DEV u16 f32_to_f16(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(u16, h);
}
// SATURATING fp16 store (+-65504, NaN stays NaN) for the UNBOUNDED forward operands: v, the GEGLU output, the normed rows (a row
// times a large gamma) and the packed model inputs.  An outlier activation of a trained checkpoint must not turn into inf -- inf * 0
// in the next GEMM would poison the whole row; a clamped outlier is a bounded error (tests/test_ops_gpu.py::test_fp16_outputs_saturate).
// q-hat / k-hat (|x| <= 8 gamma by construction) and the attention output (a convex combination of saturated v rows) use the plain
// conversion: the clamp is four VALU operations per element and measurably slowed the to_qkv epilogue when applied everywhere.
VBX_DEV u16 f32_to_f16_sat(float f) {
  f = (f > 65504.0f) ? 65504.0f : ((f < -65504.0f) ? -65504.0f : f);
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(u16, h);
}
VBX_DEV float f16_to_f32(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }
VBX_DEV unsigned pack_f16x2(float lo, float hi) { return (unsigned)f32_to_f16(lo) | ((unsigned)f32_to_f16(hi) << 16); }
VBX_DEV unsigned pack_f16x2_sat(float lo, float hi) { return (unsigned)f32_to_f16_sat(lo) | ((unsigned)f32_to_f16_sat(hi) << 16); }

// Start-phase stagger (experiment, VBX_GEMM_STAGGER=<us>): workgroups that become co-resident on a CU at launch run their
// k-loops and their epilogues in lockstep -- the matrix pipes idle while every CU stores and the memory system idles while every
// CU multiplies (tools/native/gemm_trace.cpp).  Delaying the workgroups of launch slot s (blockIdx / #CUs) by s * ticks (100 MHz
// s_memrealtime units) puts the co-residents out of phase; later workgroups inherit the phase of the slot they take over.
VBX_DEV void stagger_wait(int slot, int ticks) {
  if (slot > 0 && ticks > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long d = (unsigned long long)slot * (unsigned)ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(16);
  }
}

// ---- dropout: Philox4x32-10 (Salmon et al., SC'11), the counter-based generator torch / JAX dropout use --------------------------
// One call = 128 random bits = EIGHT 16-bit lots: element e of a call is kept iff lot_e < thr16, thr16 = round(keep * 65536)
// (<= 65535), so the kept fraction is thr16 / 65536 and the survivors are scaled by 65536 / thr16 -- exactly unbiased.
//   attention   element (bh, q, key):  counter = (4 * (key / 32) + (key % 32) / 8, q, bh, stream),  lot e = key % 8
//   feed-forward element (row, col):   counter = (col / 8, row, 0, stream),                       lot e = col % 8
// key = the 64-bit seed of this forward; stream = 2 * layer (attention) / 2 * layer + 1 (FeedForward).  The mask is a pure
// function of (seed, stream, index): backward recomputes or reloads it, nothing depends on launch geometry.
struct Philox4 { unsigned x, y, z, w; };
VBX_DEV Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// the 8 keep bits of one call (bit e = lot e kept)
VBX_DEV unsigned philox_keep8(const Philox4& r, unsigned thr16) {
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
  unsigned bits = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    bits |= ((w[i] & 0xFFFFu) < thr16 ? 1u : 0u) << (2 * i);
    bits |= ((w[i] >> 16) < thr16 ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}
static inline unsigned dropout_thr16(float p) {  // host: keep threshold of a drop probability p in (0, 1)
  long t = (long)((1.0 - (double)p) * 65536.0 + 0.5);
  return (unsigned)(t < 1 ? 1 : (t > 65535 ? 65535 : t));
}

// ---- wave64 reductions ----------------------------------------------------------------------
VBX_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
VBX_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf-GELU and its derivative -- nn.GELU()/F.gelu default, i.e. NOT the tanh approximation (voicebox_pytorch.py:217,340).
// erf itself is evaluated with Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 exact, <= 6e-7 in fp32 arithmetic): libm's
// erff costs ~40 instructions and made the FeedForward GEMM epilogue a third of its tile time; this is ~12.
VBX_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
VBX_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// the same with libm's erff (~1 ulp) for the precise mode's conv positional embedding: the A-S form's 6e-7 is ten fp32 ulps at the very
// input of a 12-layer stack.  (Round 4 measured what that is worth at the reference's chaotic initialisation: switching the FAST path's
// conv to this form moved the dim-1024 / depth-12 loss from 2.7e-3 to 5.1e-3 off the reference -- rounding-level noise, not accuracy.)
VBX_DEV float gelu_erf_libm(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
VBX_DEV float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// GEGLU packed-row map: packed row p -> reference row (or -1 when it is padding).
// Packed layout: blocks of 128 rows; first 64 = "x" rows f0..f0+63, last 64 = "gate" rows F+f0..F+f0+63.
__host__ __device__ inline int geglu_row_unmap(int p, int F) {
  int blk = p >> 7, w = p & 127;
  int f = blk * 64 + (w & 63);
  if (f >= F) return -1;
  return (w < 64) ? f : F + f;
}

// gemm3.hip: the 256 x 256 tile path (same descriptor contract as vbx_gemm; VBX_EUNSUPPORTED = "not served, use gemm.hip")
int vbx_gemm3(const vbx_gemm_desc* d, hipStream_t st);
int vbx_gemm3_tn_splitk_grouped(const vbx_gemm_desc* descs, int n, hipStream_t st);
// gemm4.hip: the 128 x 256 tile, two workgroups per CU (NT / NN descriptors)
int vbx_gemm4(const vbx_gemm_desc* d, hipStream_t st);
// gemm5.hip: the weight-stationary kernel (NT, K = 512, QKV / GEGLU epilogues)
int vbx_gemm5(const vbx_gemm_desc* d, hipStream_t st);
// 0: automatic choice per shape (default), 1: gemm.hip kernels only, 2: gemm3 wherever it can serve, 3: gemm4 wherever it can
// serve (VBX_GEMM_PATH=<n> presets it; VBX_GEMM3=0 is the same as 1)
int vbx_gemm_path();

// precise.hip: the exact-operand forward (vbx_model.precise).  runtime.hip hands over the tensors of its own arenas that the
// precise forward fills for the loss and for the (unchanged) backward; null pointers = not kept (inference).
struct VbxPreciseLayer {
  u16 *hn1, *q16, *k16, *qb, *kb, *v, *vh, *oh, *o, *hn2, *gh, *g, *h1;
  float *qrn, *krn, *lse;
  const float* b1;  // packed FeedForward[0] bias (wpack arena)
  u16* hg;                  // GateLoop: bf16 copy of its pre-norm output (the backward's wgrad operand)
  float *glp, *gls, *glh;   // GateLoop: to_qkva output, scan output, kept scan state
  unsigned *dbr, *dbc;      // attention dropout keep bits of this layer (both orientations)
};
struct VbxPreciseActs {
  float *four, *pre, *temb, *ada, *e, *pred, *per_b;
  float* const* xs;  // 2L + 1 residual snapshots
  const VbxPreciseLayer* layer;
  u16 *embed_in_bf16, *embed_in_f16, *hf;
};
int vbx_forward_precise(const vbx_model* m, const vbx_io* io, const VbxPreciseActs* a, void* stream);
