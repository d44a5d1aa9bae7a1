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
//     cross-lane traffic, the GEGLU gate sits beside its value, and a v_permlane32_swap per register pair gives 16-byte stores;
//   * one wave per SIMD means nothing but the wave's own instructions can fill the gaps of its MFMA stream: the epilogue of block j - 1 is
//     cut into 64 micro-steps that sit, pinned, behind the 64 MFMAs of block j ("Epilogues, SLOTTED" below); the MFMAs are inline asm
//     naming the weight fragments as AGPR operands (hipcc copies AGPR operands to VGPRs first: 4 v_accvgpr_read per MFMA);
//   * the phase (64 MFMAs + the slotted epilogue + the LDS-DMA pieces of block j + 2) is ONE branch-free basic block.
// Measured (tools/native/gemm5_check, B = 8 x 1040 rows, back to back): to_qkv 54-57 -> 34 us (inference outputs) / 39 us (training
// outputs), FeedForward-in 38 -> 31 / 49 -> 37 us; what each step of the way measured: docs/history.md "Round 6", DESIGN.md section 8.
// Layouts (checked on the GPU by tools/native/gemm5_check.cpp against a double-precision host reference and the 128-wide kernels):
//   MFMA 32x32x16: A operand lane l = row (l & 31), k = 8 (l >> 5) + 0..7;  B operand lane l = column (l & 31), same k;
//                  D register j of lane l = row (j & 3) + 8 (j >> 2) + 4 (l >> 5), column l & 31.
//   activation block in LDS: row t (32 rows of 1 KiB), 16-byte chunk c at chunk position c ^ (t & 15): the 16 lanes of a
//                  ds_read_b128 group (tokens {0-3,12-15,20-27} / {4-11,16-19,28-31}) hit 16 distinct bank groups.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

#ifndef VBX_G5_ABL
#define VBX_G5_ABL 0  // diagnostic builds (tools/native/g5_abl.sh): 1 no epilogue, 2 no DMA, 8 no MFMAs -- wrong results by construction
#endif
// (Measured and removed: the four waves taking turns at the texture path -- wave w issuing its pieces in slots 1 + w, 5 + w, ... --
//  is 1-3 us SLOWER than all four issuing in the same slots.  Block 0's activation rows requested before the weight rounds (weight
//  regions moved behind slot 0, 160 KiB of LDS): 39.6 / 34.4 / 38.6 / 33.1 us against 39.1 / 34.4 / 38.4 / 32.7 us in the same call --
//  nothing: the weight rounds are L2-bandwidth-bound and the first block's 32 KiB queue behind them either way.)
constexpr int G5_KS = 32;               // k-steps of 16: K = 512
constexpr int G5_K = G5_KS * 16;
constexpr int G5_ROWB = G5_K * 2;       // bytes per activation row
constexpr int G5_SLOT = 32 * G5_ROWB;   // one 32-row block
constexpr int G5_NSLOT = 3;
constexpr int G5_ROT0 = G5_NSLOT * G5_SLOT;  // rotary ring: 4 slots of 32 tokens x (cos 128 B) + 32 x (sin 128 B)
constexpr int G5_ROTSLOT = 8192;
constexpr int G5_LDS = 4 * G5_SLOT;  // 128 KiB: four wave-private 32 KiB regions while the weights load, then 96 KiB ring + 32 KiB rotary ring

struct G5Params {
  const u16* A;
  const u16* W;
  int M, nslab, npan, wpp, nrb;
  long lda, ldb;
  unsigned abytes;  // bytes of A the kernel may read: ((M - 1) lda + K) * 2 < 2^31
  int px, xs;       // XCD map: panel groups (1, 2 or 4; 0 = plain map) and workgroups per XCD
};

#ifdef VBX_G5_TRACE  // diagnostic build (tools/native/g5_trace.sh): s_memtime stamps of wave 0 of every workgroup, [wg][64] u64
__device__ unsigned long long* g5_trace_buf = nullptr;
#define G5_STAMP(i) do { if (g5_trace_buf && lane == 0 && wave == 0 && (i) < 64) g5_trace_buf[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G5_STAMP(i) do { } while (0)
#endif
__device__ uint4 g5_trash[64 * 16];
__device__ const float g5_ones[64] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                      1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f,
                                      1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};  // gamma of a v head / without qk-norm  // where the lanes of rows >= M store (256 B per lane): keeps every store unconditional

template <bool F16>
VBX_DEV f32x16 mfma32(const s16x8& a, const s16x8& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#define G5_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
// four fragments have landed: an lgkmcnt wait that their consumers depend on through the registers
template <int N, class T>
VBX_DEV void g5_wait4(T& a, T& b, T& c, T& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}

// one LDS-DMA piece (64 lanes x 16 B -> lds_dst + 16 lane) as a raw BUFFER load: descriptor {base, bytes} in SGPRs, per-lane byte
// offset voff, uniform byte offset soff; out-of-range lanes write zeros
VBX_DEV void g5_buf_lds(const void* base, unsigned bytes, char* lds_dst, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}
VBX_DEV void g5_swap(unsigned& a, unsigned& b) {  // upper half-wave of a <-> lower half-wave of b (cdna_hip_programming.md T21)
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
VBX_DEV float g5_halfsum(float x) {  // x(lane) + x(lane ^ 32) without an LDS permute
  unsigned a = __float_as_uint(x), b = a;
  g5_swap(a, b);
  return __uint_as_float(a) + __uint_as_float(b);
}
enum { G5_F16 = 0, G5_F16_SAT = 1, G5_BF16 = 2 };
template <int KIND>
VBX_DEV unsigned g5_pack(float lo, float hi) {
  if constexpr (KIND == G5_F16) return pack_f16x2(lo, hi);
  else if constexpr (KIND == G5_F16_SAT) return pack_f16x2_sat(lo, hi);
  else return pack_bf16x2(lo, hi);
}
// Two accumulator groups (8 registers = features 16 pr + {0..3, 8..11} (+4 in the upper half-wave)) of one 32-feature block as ONE
// 16-byte store per lane: after the half-wave exchange the lower lanes hold features 16 pr + 0..7, the upper ones 16 pr + 8..15.
// dst -> feature 0 of the block in this lane's row.
template <int KIND>
VBX_DEV void g5_store16(u16* dst, int pr, const float (&x)[8], int lane) {
  unsigned a0 = g5_pack<KIND>(x[0], x[1]), a1 = g5_pack<KIND>(x[2], x[3]);
  unsigned b0 = g5_pack<KIND>(x[4], x[5]), b1 = g5_pack<KIND>(x[6], x[7]);
  g5_swap(a0, b0);
  g5_swap(a1, b1);
  *reinterpret_cast<uint4*>(dst + 16 * pr + ((lane >> 5) << 3)) = make_uint4(a0, a1, b0, b1);
}
VBX_DEV void g5_load16(const float* p, int lane, float (&v)[16]) {  // v[j] = p[(j & 3) + 8 (j >> 2) + 4 (lane >> 5)]
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const float4 t = *reinterpret_cast<const float4*>(p + 8 * g + 4 * (lane >> 5));
    v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
  }
}
// (batch, token) of global row gr < 2^22: floor(gr / Np) by a float reciprocal + one correction step each way
VBX_DEV void g5_split(int gr, int Np, float inv_np, int& b, int& n) {
  b = (int)((float)gr * inv_np);
  n = gr - b * Np;
  if (n < 0) { n += Np; b--; }
  if (n >= Np) { n -= Np; b++; }
}

// ---- Epilogues, SLOTTED.  One wave per SIMD means nothing but this wave's own instructions can fill the gaps of its MFMA stream
// (a v_mfma_f32_32x32x16 occupies the matrix pipe for 32 cycles; about five other instructions issue under it), and left to the
// scheduler the two streams end up one after the other (measured: 5400 = 2930 + 2400 cycles per block).  So the epilogue of block
// j - 1 is cut into 64 micro-steps of <= ~5 vector instructions, slot<S>() for S = 0..63, and the kernel issues MFMA S of block j,
// then slot S, then a sched_barrier(0) that pins the order.  A slot reads the PREVIOUS block's accumulators (p0 / p1) and keeps its
// state in a Ctx.  No branches, every store unconditional (rows >= M go to g5_trash).  LDS reads of the epilogue (the rotary rows):
// issued by the kernel at slots 8 / 33 (reads<pr>()), complete from slots 17 / 41 on.
// The phase must stay ONE basic block: with a branch anywhere in it (as first written: around the DMA pieces of a block past the
// last) hipcc SINKS a micro-step's results towards their users in a later block and the slotting collapses into one run of vector
// instructions; pinning every result with an empty asm works too but costs a wait state per pin (~50 per phase).  G5_PIN* mark where
// the pins were (no-ops now).
#define G5_PIN1(a)
#define G5_PIN2(a, b)
#define G5_PIN4(a, b, c, d)
VBX_DEV unsigned g5_cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
VBX_DEV unsigned g5_cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
VBX_DEV float g5_clamp16(float x) {  // common.hpp f32_to_f16_sat's clamp: +-65504, NaN stays NaN (v_med3 alone would turn it into a bound)
  const float y = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
  return x != x ? x : y;
}
template <int KIND>
VBX_DEV unsigned g5_pk(float a, float b) {
  if constexpr (KIND == G5_F16) return g5_cvt_pk_f16(a, b);
  else if constexpr (KIND == G5_F16_SAT) return g5_cvt_pk_f16(g5_clamp16(a), g5_clamp16(b));
  else return g5_cvt_pk_bf16(a, b);
}
// the four packed registers of one 32-feature block's half pr (groups 2 pr, 2 pr + 1) -> one 16-byte store per lane
VBX_DEV void g5_swap4(unsigned (&k)[4]) { g5_swap(k[0], k[2]); g5_swap(k[1], k[3]); }
VBX_DEV void g5_st16(u16* dst, int pr, const unsigned (&k)[4], int lane) {
  *reinterpret_cast<uint4*>(dst + 16 * pr + ((lane >> 5) << 3)) = make_uint4(k[0], k[1], k[2], k[3]);
}

// to_qkv + MultiheadRMSNorm + rotary (the arithmetic of gemm.hip's EpiQKV on the transposed accumulators).  Slab = one head of q, k
// or v: block 0 = features d < 32, block 1 = d + 32.  KIND 0: q / k heads, KIND 1: v heads.
struct Epi5QKV {
  int Np, H;
  float qk_scale;
  const float* qg; const float* kg; const float* rc; const float* rs;
  u16* q16; u16* k16; u16* qb; u16* kb; u16* v; float* qrn; float* krn; u16* v16;
  float qps;
  float inv_np;
  static constexpr int KINDS = 2;
  static constexpr bool HAS_ROT = true;
  static constexpr bool HAS_CINIT = false;  // accumulators start at 0
  static float panel_cost(int pan, int npan) { return 3 * pan >= 2 * npan ? 0.85f : 1.0f; }  // v heads: no norm, no rotary (host side)
  struct State {
    float glo[16], ghi[16];
    int kind, head;
    u16 *d16, *db;   // q16 / k16 / v16 and qb / kb / v
    float* rn;
    float post;      // q16 = q-hat * qps
    float nscale, none;  // row multiplier = rinv * nscale + none: (qk_scale, 0), or (0, 1) without qk-norm -- arithmetic, not a branch
  };
  VBX_DEV f32x16 cinit(const State&, int) const { return f32x16{}; }  // (never used)
  struct Ctx {
    float ssa, ssb, rinv, r, rp;
    int gr, b, n;
    unsigned o;
    u16 *p16, *pb;
    float* prn;
    f32x4 e[4];
    float lo, hi, lc, hc, ol0, oh0;     // pair in flight, and the even pair's results awaiting their partner for the pack
    unsigned k16l[4], k16h[4], kbl[4], kbh[4];  // packed halves: fp16 lo / hi block, bf16 lo / hi block
  };
  template <int KIND> static constexpr int nreads() { return KIND == 0 ? 4 : 0; }
  VBX_DEV int wrow(int slab, int blk) const { return slab * 64 + blk * 32; }
  VBX_DEV void init(State& st, int slab, int lane) const {
    const int which = slab / H;
    st.head = slab - which * H;
    st.kind = which == 2 ? 1 : 0;
    st.d16 = which == 0 ? q16 : (which == 1 ? k16 : v16);
    st.db = which == 0 ? qb : (which == 1 ? kb : v);
    st.rn = which == 0 ? qrn : krn;
    st.post = which == 0 ? qps : 1.0f;
    st.nscale = qk_scale > 0.f ? qk_scale : 0.f;
    st.none = qk_scale > 0.f ? 0.f : 1.f;
    // (one unconditional load through a selected pointer: two code paths filling the arrays made hipcc keep them in scratch memory,
    //  i.e. a scratch load + vmcnt(0) in front of every use inside the MFMA phase)
    const float* g = (which < 2 && qk_scale > 0.f) ? (which == 0 ? qg : kg) + st.head * 64 : g5_ones;
    g5_load16(g, lane, st.glo);
    g5_load16(g + 32, lane, st.ghi);
  }
  // The rotary rows of a block's 32 tokens -> LDS (cos 4 KiB | sin 4 KiB; token row of 128 B, 16-byte chunk c at c ^ ((t >> 1) & 7)):
  // wave w moves tokens 8 w .. 8 w + 7, lane i = token 8 w + i / 8, chunk position i % 8.  piece 0: cos, piece 1: sin.
  VBX_DEV int rot_off(int row0, int wave, int lane, int M) const {  // this lane's byte offset into the cos / sin tables for a block
    const int tt = 8 * wave + (lane >> 3);
    int b, n;
    g5_split(max(min(row0 + tt, M - 1), 0), Np, inv_np, b, n);
    return n * 128 + ((((lane & 7) ^ (tt >> 1)) & 7) << 4);
  }
  VBX_DEV void issue_rot(int piece, char* slot, int so, int wave, bool live) const {
    g5_buf_lds(piece ? rs : rc, live ? (unsigned)(Np * 128) : 0u, slot + piece * 4096 + wave * 1024, so, 0);
  }
  // LDS reads of half pr: cos / sin of the lane's token at features 16 pr + 4 hi + {0..3}, + 8
  template <int KIND>
  VBX_DEV void reads(int pr, unsigned rot_addr, Ctx& c) const {
    if constexpr (KIND == 0) {
      if (pr == 0) {
        G5_DS_B128(c.e[0], rot_addr, 0); G5_DS_B128(c.e[1], rot_addr ^ 32, 0); G5_DS_B128(c.e[2], rot_addr, 4096); G5_DS_B128(c.e[3], rot_addr ^ 32, 4096);
      } else {
        G5_DS_B128(c.e[0], rot_addr ^ 64, 0); G5_DS_B128(c.e[1], rot_addr ^ 96, 0); G5_DS_B128(c.e[2], rot_addr ^ 64, 4096); G5_DS_B128(c.e[3], rot_addr ^ 96, 4096);
      }
    }
  }
  template <int KIND>
  VBX_DEV void wait_reads(Ctx& c, std::integral_constant<int, 0>) const { if constexpr (KIND == 0) g5_wait4<0>(c.e[0], c.e[1], c.e[2], c.e[3]); }
  template <int KIND>
  VBX_DEV void wait_reads(Ctx& c, std::integral_constant<int, 4>) const { if constexpr (KIND == 0) g5_wait4<4>(c.e[0], c.e[1], c.e[2], c.e[3]); }

  // one pair (registers J of both blocks) of a q / k head, in two parts of ~4-5 instructions
  template <bool TRAIN, int J, int PART>
  VBX_DEV void pair(const State& st, Ctx& c, const f32x16& p0, const f32x16& p1) const {
    constexpr int i = J & 7;  // index inside the half
    if constexpr (PART == 0) {
      c.lo = p0[J] * st.glo[J];
      c.hi = p1[J] * st.ghi[J];
      const float cs = c.e[i >> 2][i & 3];
      c.lc = c.lo * cs;
      c.hc = c.hi * cs;
      G5_PIN4(c.lo, c.hi, c.lc, c.hc);
    } else {
      const float sn = c.e[2 + (i >> 2)][i & 3];
      float ol = fmaf(-c.hi, sn, c.lc);  // rotate_half (voicebox_pytorch.py:193-199)
      float oh = fmaf(c.lo, sn, c.hc);
      if constexpr ((i & 1) == 0) {
        c.ol0 = ol;
        c.oh0 = oh;
        G5_PIN2(c.ol0, c.oh0);
      } else {  // the pair's partner is there: scale by the row's 1 / |x| * scale (* q prescale) and pack
        constexpr int k = i >> 1;
        if constexpr (TRAIN) {
          c.kbl[k] = g5_cvt_pk_bf16(c.ol0 * c.r, ol * c.r);
          c.kbh[k] = g5_cvt_pk_bf16(c.oh0 * c.r, oh * c.r);
        }
        c.k16l[k] = g5_cvt_pk_f16(c.ol0 * c.rp, ol * c.rp);
        c.k16h[k] = g5_cvt_pk_f16(c.oh0 * c.rp, oh * c.rp);
        G5_PIN2(c.k16l[k], c.k16h[k]);
        if constexpr (TRAIN) G5_PIN2(c.kbl[k], c.kbh[k]);
      }
    }
  }
  template <int KIND, bool TRAIN, bool F16, int S>
  VBX_DEV void slot(const State& st, Ctx& c, const f32x16& p0, const f32x16& p1, int row0, int lane, int M) const {
    u16* tr = reinterpret_cast<u16*>(g5_trash) + lane * 128;
    if constexpr (S == 11) {  // row addresses (both kinds)
      c.gr = row0 + (lane & 31);
      g5_split(max(min(c.gr, M - 1), 0), Np, inv_np, c.b, c.n);
      G5_PIN2(c.b, c.n);
    } else if constexpr (S == 12) {
      c.o = (unsigned)(((c.b * H + st.head) * Np + c.n) * 64);  // < 2^31 elements (host check)
      G5_PIN1(c.o);
    } else if constexpr (S == 13) {
      const bool valid = c.gr < M;
      c.p16 = (valid ? st.d16 : tr) + (valid ? c.o : 0u);
      if constexpr (TRAIN) c.pb = (valid ? st.db : tr) + (valid ? c.o : 0u);
    }
    if constexpr (KIND == 1) {  // v: plain head split; half pr in slots 17.. / 41..
      if constexpr ((S >= 17 && S < 21) || (S >= 41 && S < 45)) {
        constexpr int pr = S >= 41, k = (S - 17) % 24;  // packed register k of the half: accumulator registers 8 pr + 2 k, + 1
        c.k16l[k] = g5_pk<G5_F16_SAT>(p0[8 * pr + 2 * k], p0[8 * pr + 2 * k + 1]);
        c.k16h[k] = g5_pk<G5_F16_SAT>(p1[8 * pr + 2 * k], p1[8 * pr + 2 * k + 1]);
        G5_PIN2(c.k16l[k], c.k16h[k]);
        if constexpr (TRAIN) {
          c.kbl[k] = g5_cvt_pk_bf16(p0[8 * pr + 2 * k], p0[8 * pr + 2 * k + 1]);
          c.kbh[k] = g5_cvt_pk_bf16(p1[8 * pr + 2 * k], p1[8 * pr + 2 * k + 1]);
          G5_PIN2(c.kbl[k], c.kbh[k]);
        }
      }
    } else {
      if constexpr (S < 8) {  // sum of squares, two chains
        if constexpr (S == 0) { c.ssa = 0.f; c.ssb = 0.f; }
        c.ssa = fmaf(p0[2 * S], p0[2 * S], c.ssa);
        c.ssb = fmaf(p1[2 * S], p1[2 * S], c.ssb);
        c.ssa = fmaf(p0[2 * S + 1], p0[2 * S + 1], c.ssa);
        c.ssb = fmaf(p1[2 * S + 1], p1[2 * S + 1], c.ssb);
        G5_PIN2(c.ssa, c.ssb);
      } else if constexpr (S == 8) {
        c.ssa = g5_halfsum(c.ssa + c.ssb);
        G5_PIN1(c.ssa);
      } else if constexpr (S == 9) {
        // 1 / max(|x|, 1e-12) (F.normalize, voicebox_pytorch.py:286) as v_rsq + one Newton step (the raw v_rsq alone moved the chaotic
        // random-init loss, gemm.hip EpiQKV; with the step it is within an ulp of the IEEE sequence, which costs ~25 instructions)
        const float x = fmaxf(c.ssa, 1e-24f);
        const float y = __builtin_amdgcn_rsqf(x);
        c.rinv = y * fmaf(-0.5f * x * y, y, 1.5f);
        G5_PIN1(c.rinv);
      } else if constexpr (S == 10) {
        c.r = fmaf(c.rinv, st.nscale, st.none);
        c.rp = c.r * st.post;
        G5_PIN2(c.r, c.rp);
      } else if constexpr (S == 14 && TRAIN) {
        const bool wr = c.gr < M && lane < 32;
        c.prn = (wr ? st.rn : reinterpret_cast<float*>(tr)) + (wr ? (unsigned)((c.b * H + st.head) * Np + c.n) : 0u);
      } else if constexpr (S == 15 && TRAIN) {
        *c.prn = c.rinv;
      } else if constexpr (S >= 17 && S < 33) {
        this->template pair<TRAIN, ((S - 17) >> 1), ((S - 17) & 1)>(st, c, p0, p1);
      } else if constexpr (S >= 41 && S < 57) {
        this->template pair<TRAIN, (8 + ((S - 41) >> 1)), ((S - 41) & 1)>(st, c, p0, p1);
      }
    }
    // swaps and stores of a half (both kinds)
    constexpr int SW = KIND == 1 ? 21 : 33;  // first slot after half 0's packs (half 1: + 24)
    if constexpr (S == SW || S == SW + 24) {
      g5_swap4(c.k16l);
      g5_swap4(c.k16h);
    } else if constexpr (S == SW + 1 || S == SW + 25) {
      constexpr int pr = S > SW + 1;
      g5_st16(c.p16, pr, c.k16l, lane);
      g5_st16(c.p16 + 32, pr, c.k16h, lane);
    } else if constexpr (TRAIN && (S == SW + 2 || S == SW + 26)) {
      g5_swap4(c.kbl);
      g5_swap4(c.kbh);
    } else if constexpr (TRAIN && (S == SW + 3 || S == SW + 27)) {
      constexpr int pr = S > SW + 3;
      g5_st16(c.pb, pr, c.kbl, lane);
      g5_st16(c.pb + 32, pr, c.kbh, lane);
    }
  }
};

// FeedForward[0] + GEGLU.  Packed weight rows (gemm.hip EpiGEGLU): every 128 rows = 64 "x" rows then their 64 "gate" rows.
// Slab s = 32 x rows + their 32 gate rows: block 0 = x, block 1 = gate; output columns (s >> 1) * 64 + (s & 1) * 32 + 0..31.
struct Epi5GEGLU {
  u16* G; long ldg; const float* bias; u16* H1; long ldh; u16* Gb;
  static constexpr int KINDS = 1;
  static constexpr bool HAS_ROT = false;
  static constexpr bool HAS_CINIT = true;  // the bias is the C operand of a chain's first MFMA: no add in the epilogue
  static float panel_cost(int, int) { return 1.0f; }
  struct State {
    f32x16 cinit[2];  // bias of the x block / the gate block, in accumulator register order
    int kind;
    int col;  // first output column of the slab
    int wx;   // first packed weight row of the x block
  };
  VBX_DEV f32x16 cinit(const State& st, int fb) const { return st.cinit[fb]; }
  struct Ctx {
    int gr;
    unsigned og, oh;
    u16 *pg, *pgb, *ph;
    float x, g, t, pl, ee, o0, x0, g0;
    unsigned kg[4], kgb[4], khx[4], khg[4];
  };
  template <int KIND> static constexpr int nreads() { return 0; }
  VBX_DEV int wrow(int slab, int blk) const { return (slab >> 1) * 128 + blk * 64 + (slab & 1) * 32; }
  VBX_DEV void init(State& st, int slab, int lane) const {
    st.kind = 0;
    st.col = (slab >> 1) * 64 + (slab & 1) * 32;
    st.wx = wrow(slab, 0);
    float bx[16], bg[16];
    g5_load16(bias + st.wx, lane, bx);
    g5_load16(bias + st.wx + 64, lane, bg);
#pragma unroll
    for (int j = 0; j < 16; j++) { st.cinit[0][j] = bx[j]; st.cinit[1][j] = bg[j]; }
  }
  VBX_DEV int rot_off(int, int, int, int) const { return 0; }
  VBX_DEV void issue_rot(int, char*, int, int, bool) const {}
  template <int KIND> VBX_DEV void reads(int, unsigned, Ctx&) const {}
  template <int KIND, int N> VBX_DEV void wait_reads(Ctx&, std::integral_constant<int, N>) const {}
  template <bool TRAIN>
  VBX_DEV void st_half(Ctx& c, int pr, int lane) const {
    g5_st16(c.pg, pr, c.kg, lane);
    if constexpr (TRAIN) {
      g5_st16(c.pgb, pr, c.kgb, lane);
      g5_st16(c.ph, pr, c.khx, lane);
      g5_st16(c.ph + 64, pr, c.khg, lane);
    }
  }
  // The 16 outputs of a lane: output J in slots 4 J .. 4 J + 3 (half 0: slots 0-31, half 1: 32-63; the row addresses ride in the first
  // slots).  erf-GELU as common.hpp gelu_erf (Abramowitz-Stegun 7.1.26: erf z = sgn z (1 - P(t) e^{-z^2}), t = 1 / (1 + p |z|), z = g / sqrt 2)
  // with the constants folded -- t = 1 / (1 + (p / sqrt 2) |g|), e^{-z^2} = 2^(-(log2 e / 2) g^2) -- and the sign handled without a
  // select: with q = P(t) t' e^{-z^2} >= 0 and h = g / 2, gelu(g) = h (1 + sgn g (1 - q)) = (h + |h|) - |h| q.  ~15 instructions per output.
  template <int KIND, bool TRAIN, bool F16, int S>
  VBX_DEV void slot(const State& st, Ctx& c, const f32x16& p0, const f32x16& p1, int row0, int lane, int M) const {
    u16* tr = reinterpret_cast<u16*>(g5_trash) + lane * 128;
    constexpr int J = S >> 2, PART = S & 3;  // accumulator register and the quarter of its work
    if constexpr (S == 33) st_half<TRAIN>(c, 0, lane);  // half 0's stores (its swaps closed slot 31; kg[] is packed again from slot 39 on)
    if constexpr (PART == 0) {
      if constexpr (S == 0) {
        c.gr = row0 + (lane & 31);
        const int rr = max(min(c.gr, M - 1), 0);
        c.og = (unsigned)(rr * (int)ldg + st.col);
        if constexpr (TRAIN) c.oh = (unsigned)(rr * (int)ldh + st.wx);
      }
      c.x = p0[J];  // (bias included: it was the accumulator's initial value)
      c.g = p1[J];
      c.t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(c.g), 1.0f));
      c.ee = (c.g * c.g) * (-0.5f * 1.44269504088896340736f);
    } else if constexpr (PART == 1) {
      if constexpr (S == 1) {
        const bool valid = c.gr < M;
        c.pg = (valid ? G : tr) + (valid ? c.og : 0u);
        if constexpr (TRAIN) {
          c.pgb = (valid ? Gb : tr) + (valid ? c.og : 0u);
          c.ph = (valid ? H1 : tr) + (valid ? c.oh : 0u);
        }
      }
      float pl = fmaf(1.061405429f, c.t, -1.453152027f);
      pl = fmaf(pl, c.t, 1.421413741f);
      pl = fmaf(pl, c.t, -0.284496736f);
      c.pl = fmaf(pl, c.t, 0.254829592f);
      c.ee = __builtin_amdgcn_exp2f(c.ee);
    } else if constexpr (PART == 2) {
      const float q = (c.pl * c.t) * c.ee;
      const float h = 0.5f * c.g;
      c.pl = fmaf(-fabsf(h), q, h + fabsf(h)) * c.x;  // gelu(gate) * x
    } else {
      if constexpr ((J & 1) == 0) {
        c.o0 = c.pl;
        if constexpr (TRAIN) { c.x0 = c.x; c.g0 = c.g; }
      } else {
        constexpr int k = (J & 7) >> 1;
        c.kg[k] = g5_pk<F16 ? G5_F16_SAT : G5_BF16>(c.o0, c.pl);
        if constexpr (TRAIN) {
          c.kgb[k] = g5_cvt_pk_bf16(c.o0, c.pl);
          c.khx[k] = g5_cvt_pk_bf16(c.x0, c.x);
          c.khg[k] = g5_cvt_pk_bf16(c.g0, c.g);
        }
      }
      if constexpr (J == 7 || J == 15) {  // a half is packed: exchange the half-waves' registers
        g5_swap4(c.kg);
        if constexpr (TRAIN) { g5_swap4(c.kgb); g5_swap4(c.khx); g5_swap4(c.khg); }
      }
      if constexpr (J == 15) st_half<TRAIN>(c, 1, lane);
    }
  }
};

// Plain bf16 output C[M, ldc] = A . W^T (+ nothing): the K = 512 dgrads of the training step on TRANSPOSED weight copies (runtime.hip:
// d(attention out) = dx . W_out, d(GEGLU out) = dx . W_2 as NT products).  Slab s = output columns 64 s .. 64 s + 63.
struct Epi5BF16 {
  u16* C; long ldc; int N;
  static constexpr int KINDS = 1;
  static constexpr bool HAS_ROT = false;
  static constexpr bool HAS_CINIT = false;
  static float panel_cost(int, int) { return 1.0f; }
  struct State {
    int kind, col;
  };
  VBX_DEV f32x16 cinit(const State&, int) const { return f32x16{}; }
  struct Ctx {
    int gr;
    unsigned oc;
    u16* pc;
    unsigned kl[4], kh[4];
  };
  template <int KIND> static constexpr int nreads() { return 0; }
  VBX_DEV int wrow(int slab, int blk) const { return slab * 64 + blk * 32; }
  VBX_DEV void init(State& st, int slab, int) const { st.kind = 0; st.col = slab * 64; }
  VBX_DEV int rot_off(int, int, int, int) const { return 0; }
  VBX_DEV void issue_rot(int, char*, int, int, bool) const {}
  template <int KIND> VBX_DEV void reads(int, unsigned, Ctx&) const {}
  template <int KIND, int NN> VBX_DEV void wait_reads(Ctx&, std::integral_constant<int, NN>) const {}
  template <int KIND, bool TRAIN, bool F16, int S>
  VBX_DEV void slot(const State& st, Ctx& c, const f32x16& p0, const f32x16& p1, int row0, int lane, int M) const {
    u16* tr = reinterpret_cast<u16*>(g5_trash) + lane * 128;
    if constexpr (S == 0) {
      c.gr = row0 + (lane & 31);
      c.oc = (unsigned)(max(min(c.gr, M - 1), 0) * (int)ldc + st.col);
    } else if constexpr (S == 1) {
      const bool valid = c.gr < M;
      c.pc = (valid ? C : tr) + (valid ? c.oc : 0u);
    } else if constexpr ((S >= 4 && S < 8) || (S >= 36 && S < 40)) {
      constexpr int pr = S >= 36, k = S & 3;
      c.kl[k] = g5_cvt_pk_bf16(p0[8 * pr + 2 * k], p0[8 * pr + 2 * k + 1]);
      c.kh[k] = g5_cvt_pk_bf16(p1[8 * pr + 2 * k], p1[8 * pr + 2 * k + 1]);
    } else if constexpr (S == 8 || S == 40) {
      g5_swap4(c.kl);
      g5_swap4(c.kh);
    } else if constexpr (S == 9 || S == 41) {
      constexpr int pr = S == 41;
      g5_st16(c.pc, pr, c.kl, lane);
      g5_st16(c.pc + 32, pr, c.kh, lane);
    }
  }
};

template <int... I, class Fn>
VBX_DEV void g5_for_slots(std::integer_sequence<int, I...>, Fn&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <class Epi, bool F16, bool TRAIN>
__global__ __launch_bounds__(256, 1) void gemm5_kernel(G5Params p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work of this workgroup: panel `pan`, row blocks rb0, rb0 + rbs, ... (nb of them).
  //  plain map  : wpp workgroups per panel, workgroup idx of a panel takes row blocks idx, idx + wpp, ...  Every panel's workgroups
  //               sit on all 8 XCDs, so every L2 fetches the whole activation matrix (8 x the algorithmic read, PMC);
  //  XCD map    : (px > 0; launch5 picks it when it costs no extra block per workgroup) XCD x = blockIdx % 8 serves the panels of group
  //               x % px (panels g, g + px, g + 2 px, ...) and the row blocks of slice x / px only: an activation row is fetched by px
  //               L2s instead of 8.  The group's xs workgroups are dealt to its panels base + 1 to the first panels, base to the last:
  //               at to_qkv with px = 4 that is 11 / 11 / 10 for the q / k / v panel -- the v panel's blocks are the cheap ones (no
  //               norm, no rotary), so its workgroups take 13 of them while the q / k ones take 12.  Placement is a locality hint
  //               (workgroups are dealt round-robin to the XCDs), never a correctness assumption.
  int pan, rb0, rbs, nb;
  if (p.px > 0) {
    const int x = blockIdx.x & 7, s = blockIdx.x >> 3;        // XCD, slot on it (0 .. xs - 1)
    const int pg = x % p.px, rg = x / p.px, ppg = p.npan / p.px;
    const int base = p.xs / ppg, extra = p.xs - base * ppg;  // workgroups per panel of the group: base + 1 for the first `extra`
    const int cut = extra * (base + 1);
    const int pl = s < cut ? s / (base + 1) : extra + (s - cut) / base;
    const int li = s < cut ? s - pl * (base + 1) : (s - cut) - (pl - extra) * base;
    const int cnt = pl < extra ? base + 1 : base;
    const int nrg = 8 / p.px, r0 = (int)((long)rg * p.nrb / nrg), r1 = (int)((long)(rg + 1) * p.nrb / nrg);
    pan = pl * p.px + pg; rb0 = r0 + li; rbs = cnt;  // (groups interleave the panels: with 4 groups a group = one q, one k and one v panel)
    nb = rb0 < r1 ? (r1 - rb0 + cnt - 1) / cnt : 0;
  } else {
    pan = blockIdx.x / p.wpp;
    rb0 = blockIdx.x - pan * p.wpp; rbs = p.wpp;
    nb = (pan < p.npan && rb0 < p.nrb) ? (p.nrb - rb0 + p.wpp - 1) / p.wpp : 0;
  }
  if (nb == 0) return;
  const int slab_raw = pan * 4 + wave;
  const bool active = slab_raw < p.nslab;
  const int slab = active ? slab_raw : 0;  // (an idle wave of the last panel repeats slab 0 into the trash)

  // ---- the stationary weight slab: 2 feature blocks x 32 k-steps of A-operand fragments = 256 registers, ALL of the accumulator
  // half of the register file (the MFMAs below name them with "a" constraints: A / B operands may be AGPRs, so no copy is ever made).
  // Loaded THROUGH the LDS: a lane's fragments are 16-byte pieces of 32 different rows -- as direct global loads every instruction
  // touches 32 cache lines (13 us of prologue); as 1 KiB row DMAs into a wave-private 32 KiB region + the activation fragments'
  // swizzled reads it is an L2 -> LDS stream of 256 KiB per workgroup.
  s16x8 w[2][G5_KS];
  G5_STAMP(0);
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
      __builtin_amdgcn_s_barrier();  // (every wave holds its fragments before anything lands in the regions again)
    }
  }
  G5_STAMP(1);
  typename Epi::State st;
  epi.init(st, slab, lane);
  const int Meff = active ? p.M : 0;  // rows >= Meff are "invalid" for the epilogue: an idle wave stores to the trash

  auto run = [&](auto kind_c) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr int E = Epi::template nreads<KIND>();   // LDS reads of an epilogue half
    constexpr int NP = 8 + (Epi::HAS_ROT ? 2 : 0);    // LDS-DMA pieces per wave per block
    // ---- activation ring.  This wave's 8 rows of a block: t = 8 wave + q, chunk position = lane -> source chunk lane ^ (t & 15)
    // Per piece q: a constant per-lane byte offset (row 8 wave + q of the block, swizzled chunk) on a per-block uniform base, so a piece
    // costs no address arithmetic inside the phase (as first written: ~16 scalar instructions per piece, 170 per phase).  Only the
    // ragged last block (rows >= M clamp to row M - 1) computes addresses per lane.
    const int swz0 = (wave & 1) * 8;
    unsigned voff[8];
#pragma unroll
    for (int q = 0; q < 8; q++) voff[q] = (unsigned)((wave * 8 + q) * (int)p.lda * 2 + ((lane ^ (swz0 + q)) << 4));
    // X pieces as BUFFER loads to LDS (descriptor in SGPRs + constant per-lane byte offset + per-block uniform soffset: no address
    // arithmetic per piece; rows >= M are out of the buffer's range and arrive as zeros).
    struct Blk {  // what the pieces of one block share (computed once per phase, not per piece)
      unsigned xbytes;  // the X buffer's size, or 0 for a block past this workgroup's last: every lane is then out of range and the
      bool live;        // piece writes zeros into a free slot -- no branch around a piece, so the whole phase is ONE basic block
      int soff;     // byte offset of the block's first row
      char* xdst;
      char* rdst;
      int rso;      // this lane's offset into the rotary tables
    };
    auto blk_of = [&](int j) {
      const int rb = rb0 + j * rbs;
      Blk b;
      b.live = j < nb;
      b.xbytes = b.live ? p.abytes : 0u;
      b.soff = b.live ? rb * 32 * (int)p.lda * 2 : 0;
      b.xdst = smem + (j % G5_NSLOT) * G5_SLOT + wave * 8 * G5_ROWB;
      b.rdst = smem + G5_ROT0 + (j & 3) * G5_ROTSLOT;
      b.rso = Epi::HAS_ROT ? epi.rot_off(rb * 32, wave, lane, p.M) : 0;
      return b;
    };
    auto issue_piece = [&](int q, const Blk& b) {  // piece q of a block of this workgroup -> its X slot / rotary slot
      if constexpr (VBX_G5_ABL & 2) return;
      if (q < 8) {
        g5_buf_lds(p.A, b.xbytes, b.xdst + q * G5_ROWB, (int)voff[q], b.soff);
      } else {
        epi.issue_rot(q - 8, b.rdst, b.rso, wave, b.live);
      }
    };
    auto issue = [&](int j) {
      const Blk b = blk_of(j);
#pragma unroll
      for (int q = 0; q < NP; q++) issue_piece(q, b);
    };
    // fragment addresses inside a slot: token t = lane & 31, k-step s = 8 u + v: chunk 2 s + (lane >> 5) at position ^ (t & 15)
    unsigned fa[8];
    unsigned rot_a;
    {
      const int t = lane & 31;
      const unsigned a0 = (unsigned)(size_t)LDS_PTR(char, smem) + t * G5_ROWB;
#pragma unroll
      for (int v = 0; v < 8; v++) fa[v] = a0 + ((((2 * v + (lane >> 5)) ^ t) & 15) << 4);
      rot_a = (unsigned)(size_t)LDS_PTR(char, smem) + G5_ROT0 + t * 128 + ((((lane >> 5) ^ (t >> 1)) & 7) << 4);
    }
    f32x16 acc0, acc1, prv0, prv1;
    typename Epi::Ctx cx;
    // One phase = the 64 MFMAs of block j (MF), each followed by micro-step S of block j - 1's epilogue (EP) and -- in the first
    // slots -- one LDS-DMA piece of block j + 2; sched_barrier(0) after every slot pins the order.  MFMA S: k-step S / 2, feature
    // block S % 2; the fragments of k-steps 4 kb .. 4 kb + 3 are batch kb in register set kb % 2.  LDS operations in program order
    // (all inline asm) and the counted waits:  R(b0) | s0: R(b1) W(b0) | s8: R(b2) W(b1) E0 | s16: R(b3) W(b2) | s17: W(E0) |
    // s24: R(b4) W(b3) | s32: R(b5) W(b4) | s33: E1 | s40: R(b6) W(b5) | s41: W(E1) | s48: R(b7) W(b6) | s56: W(b7).
    auto phase = [&](auto mf_c, auto ep_c, int j) {
      constexpr bool MF = decltype(mf_c)::value && !(VBX_G5_ABL & 8), EP = decltype(ep_c)::value && !(VBX_G5_ABL & 1);
      constexpr int EE = EP ? E : 0, X4 = MF ? 4 : 0;
      const unsigned so = (unsigned)((j % G5_NSLOT) * G5_SLOT);
      const unsigned ra = rot_a + (unsigned)(((j - 1) & 3) * G5_ROTSLOT);
      const int row0 = (rb0 + (j - 1) * rbs) * 32;
      const Blk nxt = blk_of(j + 2);
      s16x8 xs[2][4];
#define G5_RD(kb)                                                                                       \
  G5_DS_B128(xs[(kb) & 1][0], fa[((kb) * 4 + 0) & 7] + so, (((kb) * 4 + 0) >> 3) * 256);                \
  G5_DS_B128(xs[(kb) & 1][1], fa[((kb) * 4 + 1) & 7] + so, (((kb) * 4 + 1) >> 3) * 256);                \
  G5_DS_B128(xs[(kb) & 1][2], fa[((kb) * 4 + 2) & 7] + so, (((kb) * 4 + 2) >> 3) * 256);                \
  G5_DS_B128(xs[(kb) & 1][3], fa[((kb) * 4 + 3) & 7] + so, (((kb) * 4 + 3) >> 3) * 256)
      if constexpr (MF) { G5_RD(0); }
      g5_for_slots(std::make_integer_sequence<int, 64>{}, [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr int kb = S >> 3;
        if constexpr (MF && (S & 7) == 0) {
          if constexpr (kb + 1 < 8) { G5_RD(kb + 1); }
          constexpr int N = (kb + 1 < 8 ? 4 : 0) + ((kb == 2 || kb == 5) ? EE : 0);
          g5_wait4<N>(xs[kb & 1][0], xs[kb & 1][1], xs[kb & 1][2], xs[kb & 1][3]);
        }
        if constexpr (EE > 0 && (S == 8 || S == 33)) epi.template reads<KIND>(S == 33, ra, cx);  // (33: pair 7 of half 0 still reads e[] in slot 32)
        if constexpr (EE > 0 && (S == 17 || S == 41)) epi.template wait_reads<KIND>(cx, std::integral_constant<int, X4>{});
        if constexpr (MF) {
          constexpr int ks = S >> 1, fb = S & 1;
          f32x16& acc = fb ? acc1 : acc0;
          const s16x8& x = xs[kb & 1][ks & 3];
          if constexpr (S < 2 && Epi::HAS_CINIT) {  // a chain's first MFMA: C = the epilogue's initial value (a bias), straight from its registers
            if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w[fb][ks]), "v"(x), "v"(epi.cinit(st, fb)));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w[fb][ks]), "v"(x), "v"(epi.cinit(st, fb)));
          } else if constexpr (S < 2) {
            if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w[fb][ks]), "v"(x));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w[fb][ks]), "v"(x));
          } else {
            if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w[fb][ks]), "v"(x));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w[fb][ks]), "v"(x));
          }
        }
        // the LDS-DMA pieces of block j + 2 (after this phase's barrier: their slots are free), early in the phase so that they have
        // landed by the next one's vmcnt(NP) -- that allowance is then what lets this phase's STORES stay in flight
        if constexpr (decltype(mf_c)::value && (S & 1) == 1 && (S >> 1) < NP) issue_piece(S >> 1, nxt);
        if constexpr (EP) epi.template slot<KIND, TRAIN, F16, S>(st, cx, prv0, prv1, row0, lane, Meff);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (MF) {
        // an MFMA's result may be read by a vector instruction only 18 + wait states after a 16-pass MFMA issued (the assembler does
        // not pad inline asm): the copies below are the first readers
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc0), "+v"(acc1));
        prv0 = acc0;
        prv1 = acc1;
      }
    };
    // block j has landed (this wave's pieces; the barrier makes it everyone's).  At most the NP pieces of block j + 1 may stay in
    // flight: loads return in order, so <= NP outstanding operations of any kind means block j is complete -- and block j + 1 was
    // requested early in the previous phase, so in practice the allowance is what lets the previous epilogue's STORES stay in flight.
    auto top = [&](int j) {
      G5_STAMP(2 + 4 * j);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");  // (every phase issues NP pieces, the last two into free slots with nothing to read)
      G5_STAMP(3 + 4 * j);
      __builtin_amdgcn_s_barrier();  // ... and every wave is done reading block j - 1 (X slot) and j - 2 (rotary slot): they take block j + 2
      G5_STAMP(4 + 4 * j);
      G5_STAMP(5 + 4 * j);
    };
    using T = std::true_type;
    using F = std::false_type;
    issue(0);
    issue(1);
    top(0);
    phase(T{}, F{}, 0);
    for (int j = 1; j < nb; j++) {
      top(j);
      phase(T{}, T{}, j);
    }
    G5_STAMP(2 + 4 * nb);
    phase(F{}, T{}, nb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA piece may still be in flight when the workgroup's LDS is handed on
    G5_STAMP(3 + 4 * nb);
  };
  if constexpr (Epi::KINDS == 2) {
    if (st.kind == 1) run(std::integral_constant<int, 1>{});
    else run(std::integral_constant<int, 0>{});
  } else {
    run(std::integral_constant<int, 0>{});
  }
}

template <class Epi, bool F16, bool TRAIN>
int launch5k(const G5Params& p, const Epi& epi, int grid, hipStream_t st) {
  static bool attr = false;
  auto k = gemm5_kernel<Epi, F16, TRAIN>;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G5_LDS); attr = true; }
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), G5_LDS, st, p, epi);
  VBX_LAUNCH_CHECK();
  return 0;
}
int g5_cu_limit = 0;  // vbx_gemm5_cu_limit
template <class Epi>
int launch5(const vbx_gemm_desc* d, const Epi& epi, int nslab, bool train, hipStream_t st) {
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
  if (((long)d->M + 32) * d->lda * 2 >= (1L << 31)) return VBX_EUNSUPPORTED;
  p.abytes = (unsigned)((((long)d->M - 1) * d->lda + G5_K) * 2);
  const int cus = (g5_cu_limit > 0 && g5_cu_limit < ncu) ? g5_cu_limit : ncu;
  if (p.npan > cus || d->M >= (1 << 22)) return VBX_EUNSUPPORTED;
  p.wpp = cus / p.npan;
  if (p.wpp > p.nrb) p.wpp = p.nrb;
  int grid = p.npan * p.wpp;
  p.px = 0; p.xs = 0;
  static const bool xcd_map = !(getenv("VBX_GEMM5_XCD") && atoi(getenv("VBX_GEMM5_XCD")) == 0);
  if (xcd_map && cus == ncu && ncu % 8 == 0 && p.nrb >= 8 * 8) {
    // the XCD map with the lowest cost = max over panels of (row blocks per workgroup x the panel's relative block time, Epi::panel_cost:
    // a v head of to_qkv has a light epilogue), if that is no worse than the plain map; ties go to fewer panel groups (= fewer L2s
    // fetching an activation row).  VBX_GEMM5_PX=<1|2|4> forces one (A/B).
    static const int px_force = getenv("VBX_GEMM5_PX") ? atoi(getenv("VBX_GEMM5_PX")) : 0;
    const int xs = ncu / 8;
    float best = (float)cdiv(p.nrb, p.wpp);
    for (int px = 1; px <= 4; px *= 2) {
      if (p.npan % px || (px_force && px != px_force)) continue;
      const int ppg = p.npan / px, nrg = 8 / px;
      if (ppg > xs) continue;
      const int base = xs / ppg, extra = xs - base * ppg, rows_max = cdiv(p.nrb, nrg);  // (slices differ by at most one row block)
      float cost = 0.f;
      for (int pl = 0; pl < ppg; pl++)
        for (int pg = 0; pg < px; pg++) {
          const float c = (float)cdiv(rows_max, pl < extra ? base + 1 : base) * Epi::panel_cost(pl * px + pg, p.npan);
          cost = c > cost ? c : cost;
        }
      if (cost < best - 1e-3f || (px_force && !p.px) || (!p.px && cost <= best + 1e-3f)) { best = cost < best ? cost : best; p.px = px; p.xs = xs; grid = ncu; }
    }
  }
  if (d->f16) return train ? launch5k<Epi, true, true>(p, epi, grid, st) : launch5k<Epi, true, false>(p, epi, grid, st);
  return train ? launch5k<Epi, false, true>(p, epi, grid, st) : launch5k<Epi, false, false>(p, epi, grid, st);
}

}  // namespace

extern "C" int vbx_gemm5_cu_limit(int n) {
  VBX_REQUIRE(n >= 0, "vbx_gemm5_cu_limit: n >= 0 (0 = every CU)");
  g5_cu_limit = n;
  return 0;
}
#ifdef VBX_G5_TRACE
extern "C" int vbx_debug_gemm5_trace(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g5_trace_buf), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#endif
// Serves NT descriptors with K = 512 and the QKV / GEGLU epilogues; VBX_EUNSUPPORTED = "not mine" (vbx_gemm then uses the LDS-tiled kernels).
int vbx_gemm5(const vbx_gemm_desc* d, hipStream_t st) {
  if (d->mode != VBX_GEMM_NT || d->K != G5_K || d->lda % 8 || d->ldb % 8 || d->M < 1) return VBX_EUNSUPPORTED;
  if ((reinterpret_cast<size_t>(d->A) | reinterpret_cast<size_t>(d->B)) & 15) return VBX_EUNSUPPORTED;
  if (d->epilogue == VBX_EPI_QKV) {
    if (!(d->H > 0 && d->N == 3 * d->H * 64 && d->Np > 0 && d->M % d->Np == 0)) return VBX_EUNSUPPORTED;
    if (!(d->q16 && d->k16 && d->v16 && d->rot_cos && d->rot_sin)) return VBX_EUNSUPPORTED;
    if (d->qk_scale > 0.f && !(d->q_gamma && d->k_gamma)) return VBX_EUNSUPPORTED;
    if ((long)d->M * d->H * 64 >= (1L << 31)) return VBX_EUNSUPPORTED;
    const int ntrain = (d->qb != nullptr) + (d->kb != nullptr) + (d->v != nullptr) + (d->q_rnorm != nullptr) + (d->k_rnorm != nullptr);
    if (ntrain != 0 && ntrain != 5) return VBX_EUNSUPPORTED;  // all of the backward's copies or none
    Epi5QKV e{d->Np, d->H, d->qk_scale, d->q_gamma, d->k_gamma, d->rot_cos, d->rot_sin,
              (u16*)d->q16, (u16*)d->k16, (u16*)d->qb, (u16*)d->kb, (u16*)d->v, d->q_rnorm, d->k_rnorm, (u16*)d->v16,
              d->q_prescale > 0.f ? d->q_prescale : 1.0f, 1.0f / (float)d->Np};
    return launch5(d, e, d->N / 64, ntrain == 5, st);
  }
  if (d->epilogue == VBX_EPI_GEGLU) {
    if (d->N % 128 || !d->bias || !d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
    if ((d->C2 != nullptr) != (d->C3 != nullptr)) return VBX_EUNSUPPORTED;
    if ((long)d->M * d->N >= (1L << 31) || (long)d->M * d->ldc >= (1L << 31)) return VBX_EUNSUPPORTED;
    Epi5GEGLU e{(u16*)d->C, d->ldc, d->bias, (u16*)d->C2, d->N, (u16*)d->C3};
    return launch5(d, e, d->N / 64, d->C2 != nullptr, st);
  }
  if (d->epilogue == VBX_EPI_BF16) {
    if (d->bias || !d->C || d->ldc % 8 || d->N % 64 || d->delta || d->f16) return VBX_EUNSUPPORTED;
    if ((long)d->M * d->ldc >= (1L << 31)) return VBX_EUNSUPPORTED;
    Epi5BF16 e{(u16*)d->C, d->ldc, d->N};
    return launch5(d, e, cdiv(d->N, 64), false, st);
  }
  return VBX_EUNSUPPORTED;
}
