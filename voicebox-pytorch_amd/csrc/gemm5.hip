// gemm5: WEIGHT-STATIONARY GEMM for the model's wide, SHORT-K linear layers with a row-wise epilogue (to_qkv + MultiheadRMSNorm +
// rotary, FeedForward-in + GEGLU: voicebox_pytorch.py:320-328, 338-345) at K = dim = 512, gfx950 only.
//
// Why another tile.  Every LDS-tiled form of these GEMMs (128 x 128 three per CU, 128 x 256 two per CU, 256 x 256, 64-deep one-round)
// streams BOTH operands L2 -> LDS per k-tile and lands at 45-54 us: the L2 -> LDS operand stream (27-52 B/clk per CU) caps the k-loop at
// ~800 TFLOP/s, and a VALU epilogue of the same order as the 12-14 us k-loop follows every tile (docs/history.md).  With K = 512 the whole
// weight slab of 64 output features is 64 KiB = 256 registers of a wave: so the WEIGHTS sit in registers for the life of the
// workgroup (the unified 512-entry file of gfx950, one wave per SIMD), and the only operand that moves is the activation block:
//   * a workgroup = 4 waves = a PANEL of 4 x 64 output features; it walks 32-row blocks of the activations (its share of M);
//   * per block 32 KiB of activations arrive by LDS-DMA (one 1 KiB row per instruction, 3-slot ring, one barrier per block) and every
//     wave reads them once as the MFMA "B" operand: 32 ds_read_b128 per 64 v_mfma_f32_32x32x16 -- a quarter of the LDS reads and a
//     sixth of the L2 -> LDS bytes per MFMA of the 128 x 128 tile;
//   * the product is computed TRANSPOSED (features x tokens): a lane then owns ONE token and, of its 64 features, the 16 rotary
//     pairs (d, d + 32) -- acc0[j] / acc1[j] -- so the sum of squares is 32 FMAs + one half-wave exchange, rotate_half needs no
//     cross-lane traffic, the GEGLU gate sits beside its value, and a v_permlane32_swap per register pair gives 16-byte stores.
// Layouts (checked on the GPU by tools/native/gemm5_check.cpp against a double-precision host reference and the 128-wide kernels):
//   MFMA 32x32x16: A operand lane l = row (l & 31), k = 8 (l >> 5) + 0..7;  B operand lane l = column (l & 31), same k;
//                  D register j of lane l = row (j & 3) + 8 (j >> 2) + 4 (l >> 5), column l & 31.
//   activation block in LDS: row t (32 rows of 1 KiB), 16-byte chunk c at chunk position c ^ (t & 15): the 16 lanes of a
//                  ds_read_b128 group (tokens {0-3,12-15,20-27} / {4-11,16-19,28-31}) hit 16 distinct bank groups.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int G5_KS = 32;               // k-steps of 16: K = 512
constexpr int G5_K = G5_KS * 16;
constexpr int G5_ROWB = G5_K * 2;       // bytes per activation row
constexpr int G5_SLOT = 32 * G5_ROWB;   // one 32-row block
constexpr int G5_NSLOT = 3;
constexpr int G5_LDS = 4 * G5_SLOT;  // 128 KiB: four wave-private 32 KiB regions while the weights load, then the 3-slot ring

struct G5Params {
  const u16* A;
  const u16* W;
  int M, nslab, npan, wpp, nrb;
  long lda, ldb;
  int abl;  // timing ablations (VBX_G5_ABL, tools only; wrong results): 1 no epilogue, 2 no DMA, 4 every workgroup reads row block 0, 8 no MFMAs
};

template <bool F16>
VBX_DEV f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#define G5_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
// the four fragments of a batch have landed: an lgkmcnt wait that the MFMAs depend on through the registers (no sched_barrier needed)
template <int N>
VBX_DEV void g5_wait4(s16x8& a, s16x8& b, s16x8& c, s16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}

// feature of accumulator register j inside its 32-feature block
VBX_DEV int g5_feat(int j, int lane) { return (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5); }

VBX_DEV void g5_swap(unsigned& a, unsigned& b) {  // upper half-wave of a <-> lower half-wave of b (cdna_hip_programming.md T21)
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
enum { G5_F16 = 0, G5_F16_SAT = 1, G5_BF16 = 2 };
template <int KIND>
VBX_DEV unsigned g5_pack(float lo, float hi) {
  if constexpr (KIND == G5_F16) return pack_f16x2(lo, hi);
  else if constexpr (KIND == G5_F16_SAT) return pack_f16x2_sat(lo, hi);
  else return pack_bf16x2(lo, hi);
}
// 32 features of one token (the lane's 16 + its partner half-wave's 16) as two 16-byte stores per lane: dst -> feature 0 of the block
template <int KIND>
VBX_DEV void g5_store32(u16* dst, const float (&x)[16], int lane, bool valid) {
#pragma unroll
  for (int gp = 0; gp < 4; gp += 2) {
    unsigned a0 = g5_pack<KIND>(x[4 * gp + 0], x[4 * gp + 1]), a1 = g5_pack<KIND>(x[4 * gp + 2], x[4 * gp + 3]);
    unsigned b0 = g5_pack<KIND>(x[4 * gp + 4], x[4 * gp + 5]), b1 = g5_pack<KIND>(x[4 * gp + 6], x[4 * gp + 7]);
    g5_swap(a0, b0);
    g5_swap(a1, b1);
    if (valid) *reinterpret_cast<uint4*>(dst + 8 * gp + ((lane >> 5) << 3)) = make_uint4(a0, a1, b0, b1);
  }
}
VBX_DEV void g5_load16(const float* p, int lane, float (&v)[16]) {  // v[j] = p[g5_feat(j, lane)]
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float4 t = *reinterpret_cast<const float4*>(p + 8 * g + 4 * (lane >> 5));
    v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
  }
}

// ---- to_qkv + MultiheadRMSNorm + rotary (the arithmetic of gemm.hip's EpiQKV on the transposed accumulators).  Slab = one head of
// q, k or v: block 0 = features d < 32, block 1 = d + 32.
struct Epi5QKV {
  int Np, H;
  float qk_scale;
  const float* qg; const float* kg; const float* rc; const float* rs;
  u16* q16; u16* k16; u16* qb; u16* kb; u16* v; float* qrn; float* krn; u16* v16;
  float qps;
  struct State {
    float glo[16], ghi[16];
    int which, head;
  };
  VBX_DEV int wrow(int slab, int blk) const { return slab * 64 + blk * 32; }
  VBX_DEV void init(State& st, int slab, int lane) const {
    st.which = slab / H;
    st.head = slab - st.which * H;
    if (st.which < 2 && qk_scale > 0.f) {
      const float* g = (st.which == 0 ? qg : kg) + st.head * 64;
      g5_load16(g, lane, st.glo);
      g5_load16(g + 32, lane, st.ghi);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) st.glo[j] = st.ghi[j] = 1.f;
    }
  }
  VBX_DEV void operator()(const State& st, const f32x16& a0, const f32x16& a1, int row0, int lane, int M) const {
    const int gr = row0 + (lane & 31);
    const bool valid = gr < M;
    const int grc = valid ? gr : M - 1;
    const int b = grc / Np, n = grc - b * Np;
    const long o = (((long)b * H + st.head) * Np + n) * 64;
    float lo[16], hi[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { lo[j] = a0[j]; hi[j] = a1[j]; }
    if (st.which == 2) {  // v: plain head split
      if (v) { g5_store32<G5_BF16>(v + o, lo, lane, valid); g5_store32<G5_BF16>(v + o + 32, hi, lane, valid); }
      if (v16) { g5_store32<G5_F16_SAT>(v16 + o, lo, lane, valid); g5_store32<G5_F16_SAT>(v16 + o + 32, hi, lane, valid); }
      return;
    }
    float cs[16], sn[16];
    g5_load16(rc + (long)n * 32, lane, cs);
    g5_load16(rs + (long)n * 32, lane, sn);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) ss = fmaf(lo[j], lo[j], fmaf(hi[j], hi[j], ss));
    ss += __shfl_xor(ss, 32, 64);
    const float rinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize (voicebox_pytorch.py:286); the IEEE sequence, as EpiQKV
    const bool isq = st.which == 0;
    if (qk_scale > 0.f) {
      const float r = rinv * qk_scale;
#pragma unroll
      for (int j = 0; j < 16; j++) { lo[j] = lo[j] * r * st.glo[j]; hi[j] = hi[j] * r * st.ghi[j]; }
    }
    // rotate_half (voicebox_pytorch.py:193-199)
    float olo[16], ohi[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      olo[j] = lo[j] * cs[j] - hi[j] * sn[j];
      ohi[j] = hi[j] * cs[j] + lo[j] * sn[j];
    }
    u16* bcopy = isq ? qb : kb;
    if (bcopy) { g5_store32<G5_BF16>(bcopy + o, olo, lane, valid); g5_store32<G5_BF16>(bcopy + o + 32, ohi, lane, valid); }
    if (isq) {
#pragma unroll
      for (int j = 0; j < 16; j++) { olo[j] *= qps; ohi[j] *= qps; }
    }
    u16* dst = isq ? q16 : k16;
    g5_store32<G5_F16>(dst + o, olo, lane, valid);
    g5_store32<G5_F16>(dst + o + 32, ohi, lane, valid);
    float* rn = isq ? qrn : krn;
    if (rn && valid && lane < 32) rn[((long)b * H + st.head) * Np + n] = rinv;
  }
};

// ---- FeedForward[0] + GEGLU.  Packed weight rows (gemm.hip EpiGEGLU): every 128 rows = 64 "x" rows then their 64 "gate" rows.
// Slab s = 32 x rows + their 32 gate rows: block 0 = x, block 1 = gate; output columns (s >> 1) * 64 + (s & 1) * 32 + 0..31.
struct Epi5GEGLU {
  u16* G; long ldg; const float* bias; u16* H1; long ldh; u16* Gb; int g_f16;
  struct State {
    float bx[16], bg[16];
    int col;  // first output column of the slab
    int wx;   // first packed weight row of the x block
  };
  VBX_DEV int wrow(int slab, int blk) const { return (slab >> 1) * 128 + blk * 64 + (slab & 1) * 32; }
  VBX_DEV void init(State& st, int slab, int lane) const {
    st.col = (slab >> 1) * 64 + (slab & 1) * 32;
    st.wx = wrow(slab, 0);
    g5_load16(bias + st.wx, lane, st.bx);
    g5_load16(bias + st.wx + 64, lane, st.bg);
  }
  VBX_DEV void operator()(const State& st, const f32x16& a0, const f32x16& a1, int row0, int lane, int M) const {
    const int gr = row0 + (lane & 31);
    const bool valid = gr < M;
    float x[16], g[16], o[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      x[j] = a0[j] + st.bx[j];
      g[j] = a1[j] + st.bg[j];
      o[j] = gelu_erf(g[j]) * x[j];
    }
    u16* gdst = G + (long)gr * ldg + st.col;
    if (g_f16) g5_store32<G5_F16_SAT>(gdst, o, lane, valid);
    else g5_store32<G5_BF16>(gdst, o, lane, valid);
    if (Gb) g5_store32<G5_BF16>(Gb + (long)gr * ldg + st.col, o, lane, valid);
    if (H1) {
      g5_store32<G5_BF16>(H1 + (long)gr * ldh + st.wx, x, lane, valid);
      g5_store32<G5_BF16>(H1 + (long)gr * ldh + st.wx + 64, g, lane, valid);
    }
  }
};

template <class Epi, bool F16>
__global__ __launch_bounds__(256, 1) void gemm5_kernel(G5Params p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pan = blockIdx.x / p.wpp, idx = blockIdx.x - pan * p.wpp;
  if (pan >= p.npan || idx >= p.nrb) return;
  const int nb = (p.nrb - idx + p.wpp - 1) / p.wpp;  // this workgroup's blocks: idx, idx + wpp, ...
  const int slab_raw = pan * 4 + wave;
  const bool active = slab_raw < p.nslab;
  const int slab = active ? slab_raw : 0;

  // ---- the stationary weight slab: 2 feature blocks x 32 k-steps of A-operand fragments (256 registers).  Loaded THROUGH the LDS:
  // a lane's fragments are 16-byte pieces of 32 different rows -- as direct global loads every instruction touches 32 cache lines
  // (13 us of prologue, VBX_G5_ABL=11); as 1 KiB row DMAs into a wave-private 32 KiB region + the activation fragments' swizzled
  // ds_read_b128 it is an L2 -> LDS stream of 256 KiB per workgroup.
  s16x8 w[2][G5_KS];
  {
    char* reg = smem + wave * G5_SLOT;
    const int t = lane & 31;
#pragma unroll
    for (int b = 0; b < 2; b++) {
      const u16* wr = p.W + (long)epi.wrow(slab, b) * p.ldb;
#pragma unroll
      for (int r = 0; r < 32; r++) {
        const u16* src = wr + (long)r * p.ldb + ((lane ^ (r & 15)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(reg + r * G5_ROWB), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // LDS-DMA data is ordered for a ds_read only by the counted vmcnt FOLLOWED BY A BARRIER (even for the issuing wave)
      // compiler-visible LDS loads: hipcc tracks their lgkmcnt itself (an inline-asm read + a later wait would leave a window in
      // which the allocator may copy a register whose data has not landed -- with 256 live values it does)
#pragma unroll
      for (int s = 0; s < G5_KS; s++) {
        const int off = t * G5_ROWB + ((s >> 3) << 8) + ((((2 * (s & 7) + (lane >> 5)) ^ t) & 15) << 4);
        w[b][s] = *LDS_PTR(const s16x8, reg + off);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // (every wave holds its fragments before anything lands in the region again)
    }
    // (the regions become the activation ring)
  }
  typename Epi::State st;
  epi.init(st, slab, lane);

  // ---- activation ring.  This wave's 8 rows of a block: t = 8 wave + q, chunk position = lane -> source chunk lane ^ (t & 15)
  const int swz0 = (wave & 1) * 8;
  auto issue = [&](int j) {  // block j of this workgroup -> slot j % 3
    if (p.abl & 2) return;
    const int rb = (p.abl & 4) ? 0 : idx + j * p.wpp;
    char* dst = smem + (j % G5_NSLOT) * G5_SLOT + wave * 8 * G5_ROWB;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int row = min(rb * 32 + wave * 8 + q, p.M - 1);
      const u16* src = p.A + (long)row * p.lda + ((lane ^ (swz0 + q)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + q * G5_ROWB), 16, 0, 0);
    }
  };
  // fragment addresses inside a slot: token t = lane & 31, k-step s = 8 u + v: chunk 2 s + (lane >> 5) at position ^ (t & 15)
  unsigned fa[8];
  {
    const int t = lane & 31;
    const unsigned a0 = (unsigned)(size_t)LDS_PTR(char, smem) + t * G5_ROWB;
#pragma unroll
    for (int v = 0; v < 8; v++) fa[v] = a0 + ((((2 * v + (lane >> 5)) ^ t) & 15) << 4);
  }

  issue(0);
  if (nb > 1) issue(1);
  for (int j = 0; j < nb; j++) {
    // block j has landed (this wave's pieces; the barrier makes it everyone's).  At most the 8 pieces of block j + 1 may stay in
    // flight: loads return in order, so <= 8 outstanding operations of any kind means block j is complete.
    if (j + 1 < nb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // ... and every wave is done reading block j - 1: its slot takes block j + 2
    if (j + 2 < nb) issue(j + 2);
    const unsigned so = (unsigned)((j % G5_NSLOT) * G5_SLOT);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; i++) acc0[i] = acc1[i] = 0.f;
    s16x8 xa[4], xb[4];
#define G5_READ4(x, kb)                                                     \
  G5_DS_B128(x[0], fa[((kb) * 4 + 0) & 7] + so, (((kb) * 4 + 0) >> 3) * 256); \
  G5_DS_B128(x[1], fa[((kb) * 4 + 1) & 7] + so, (((kb) * 4 + 1) >> 3) * 256); \
  G5_DS_B128(x[2], fa[((kb) * 4 + 2) & 7] + so, (((kb) * 4 + 2) >> 3) * 256); \
  G5_DS_B128(x[3], fa[((kb) * 4 + 3) & 7] + so, (((kb) * 4 + 3) >> 3) * 256)
#define G5_MFMA4(x, kb)                                        \
  _Pragma("unroll") for (int i = 0; i < 4; i++) {              \
    acc0 = mfma32<F16>(w[0][(kb) * 4 + i], x[i], acc0);        \
    acc1 = mfma32<F16>(w[1][(kb) * 4 + i], x[i], acc1);        \
  }
    if (!(p.abl & 8)) {
    G5_READ4(xa, 0);
#pragma unroll
    for (int kb = 0; kb < 8; kb += 2) {
      G5_READ4(xb, kb + 1);
      g5_wait4<4>(xa[0], xa[1], xa[2], xa[3]);
      G5_MFMA4(xa, kb);
      if (kb + 2 < 8) {
        G5_READ4(xa, kb + 2);
        g5_wait4<4>(xb[0], xb[1], xb[2], xb[3]);
      } else {
        g5_wait4<0>(xb[0], xb[1], xb[2], xb[3]);
      }
      G5_MFMA4(xb, kb + 1);
    }
    }
#undef G5_READ4
#undef G5_MFMA4
    if (active && !(p.abl & 1)) epi(st, acc0, acc1, (idx + j * p.wpp) * 32, lane, p.M);
    if (p.abl & 1) { if (acc0[0] + acc1[3] == 123.456f) *(float*)smem = acc0[5]; }
  }
}

template <class Epi>
int launch5(const vbx_gemm_desc* d, const Epi& epi, int nslab, hipStream_t st) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VBX_EUNSUPPORTED;
    ncu = prop.multiProcessorCount;
  }
  G5Params p;
  p.A = (const u16*)d->A; p.W = (const u16*)d->B; p.M = d->M; p.lda = d->lda; p.ldb = d->ldb;
  p.nslab = nslab; p.npan = cdiv(nslab, 4); p.nrb = cdiv(d->M, 32);
  if (p.npan > ncu) return VBX_EUNSUPPORTED;
  p.wpp = ncu / p.npan;
  if (p.wpp > p.nrb) p.wpp = p.nrb;
  const int grid = p.npan * p.wpp;
  static const int abl = getenv("VBX_G5_ABL") ? atoi(getenv("VBX_G5_ABL")) : 0;
  p.abl = abl;
  if (d->f16) {
    static bool attr = false;
    auto k = gemm5_kernel<Epi, true>;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS); attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), G5_LDS, st, p, epi);
  } else {
    static bool attr = false;
    auto k = gemm5_kernel<Epi, false>;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS); attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), G5_LDS, st, p, epi);
  }
  VBX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Serves NT descriptors with K = 512 and the QKV / GEGLU epilogues; VBX_EUNSUPPORTED = "not mine" (vbx_gemm then uses the LDS-tiled kernels).
int vbx_gemm5(const vbx_gemm_desc* d, hipStream_t st) {
  if (d->mode != VBX_GEMM_NT || d->K != G5_K || d->lda % 8 || d->ldb % 8 || d->M < 1) return VBX_EUNSUPPORTED;
  if ((reinterpret_cast<size_t>(d->A) | reinterpret_cast<size_t>(d->B)) & 15) return VBX_EUNSUPPORTED;
  if (d->epilogue == VBX_EPI_QKV) {
    if (!(d->H > 0 && d->N == 3 * d->H * 64 && d->Np > 0 && d->M % d->Np == 0)) return VBX_EUNSUPPORTED;
    if (!(d->q16 && d->k16 && (d->v || d->v16) && d->rot_cos && d->rot_sin)) return VBX_EUNSUPPORTED;
    if (d->qk_scale > 0.f && !(d->q_gamma && d->k_gamma)) return VBX_EUNSUPPORTED;
    Epi5QKV e{d->Np, d->H, d->qk_scale, d->q_gamma, d->k_gamma, d->rot_cos, d->rot_sin,
              (u16*)d->q16, (u16*)d->k16, (u16*)d->qb, (u16*)d->kb, (u16*)d->v, d->q_rnorm, d->k_rnorm, (u16*)d->v16,
              d->q_prescale > 0.f ? d->q_prescale : 1.0f};
    return launch5(d, e, d->N / 64, st);
  }
  if (d->epilogue == VBX_EPI_GEGLU) {
    if (d->N % 128 || !d->bias || !d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
    Epi5GEGLU e{(u16*)d->C, d->ldc, d->bias, (u16*)d->C2, d->N, (u16*)d->C3, d->f16};
    return launch5(d, e, d->N / 64, st);
  }
  return VBX_EUNSUPPORTED;
}
