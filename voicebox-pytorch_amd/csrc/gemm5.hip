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

#ifndef VBX_G5_ABL
#define VBX_G5_ABL 0  // diagnostic builds (tools/native/g5_abl.sh): 1 no epilogue, 2 no DMA, 8 no MFMAs -- wrong results by construction
#endif
#ifndef G5_VALU_PER_MFMA
#define G5_VALU_PER_MFMA 5
#endif
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
};

#ifdef VBX_G5_TRACE  // diagnostic build (tools/native/g5_trace.sh): s_memtime stamps of wave 0 of every workgroup, [wg][64] u64
__device__ unsigned long long* g5_trace_buf = nullptr;
#define G5_STAMP(i) do { if (g5_trace_buf && lane == 0 && wave == 0 && (i) < 64) g5_trace_buf[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G5_STAMP(i) do { } while (0)
#endif
__device__ uint4 g5_trash[64 * 16];  // where the lanes of rows >= M store (256 B per lane): keeps every store unconditional

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

// ---- Epilogues.  A workgroup's waves run the epilogue of block j - 1 INSIDE the MFMA phase of block j (one basic block: no
// branches, every store unconditional -- rows >= M go to g5_trash), in three pieces: pre (row statistics, addresses), then the two
// halves pr = 0, 1 of the accumulator registers (8 registers of each feature block), each with up to four 16-byte LDS reads
// (nreads) that the kernel places between its own fragment reads and waits for by count.
//
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
  struct State {
    float glo[16], ghi[16];
    int kind, head;
    u16 *d16, *db;   // q16 / k16 / v16 and qb / kb / v
    float* rn;
    float post;      // q16 = q-hat * qps
    float nscale, none;  // row multiplier = rinv * nscale + none: (qk_scale, 0), or (0, 1) without qk-norm -- arithmetic, not a branch
  };
  struct Row {
    u16 *p16, *pb;
    float r;
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
    if (which < 2 && qk_scale > 0.f) {
      const float* g = (which == 0 ? qg : kg) + st.head * 64;
      g5_load16(g, lane, st.glo);
      g5_load16(g + 32, lane, st.ghi);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) st.glo[j] = st.ghi[j] = 1.f;
    }
  }
  // the rotary rows of a block's 32 tokens -> LDS (cos 4 KiB | sin 4 KiB; token row of 128 B, 16-byte chunk c at c ^ ((t >> 1) & 7)):
  // wave w moves tokens 8 w .. 8 w + 7, lane i = token 8 w + i / 8, chunk position i % 8
  VBX_DEV void issue_rot(char* slot, int row0, int wave, int lane, int M) const {
    const int tt = 8 * wave + (lane >> 3);
    int b, n;
    g5_split(min(row0 + tt, M - 1), Np, inv_np, b, n);
    const long so = (long)n * 32 + ((((lane & 7) ^ (tt >> 1)) & 7) << 2);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rc + so),
                                     (__attribute__((address_space(3))) void*)(slot + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rs + so),
                                     (__attribute__((address_space(3))) void*)(slot + 4096 + wave * 1024), 16, 0, 0);
  }
  template <int KIND, bool TRAIN>
  VBX_DEV void pre(const State& st, Row& rw, const f32x16& a0, const f32x16& a1, int row0, int lane, int M) const {
    const int gr = row0 + (lane & 31);
    const bool valid = gr < M;
    int b, n;
    g5_split(max(min(gr, M - 1), 0), Np, inv_np, b, n);  // (always a real row: nothing below is worth a branch to the compiler)
    unsigned o = (unsigned)(((b * H + st.head) * Np + n) * 64);  // < 2^31 elements (host check)
    asm volatile("" : "+v"(o));  // (computed for every lane: hipcc otherwise wraps it in an exec branch and splits the MFMA block)
    u16* tr = reinterpret_cast<u16*>(g5_trash) + lane * 128;
    rw.p16 = (valid ? st.d16 : tr) + (valid ? o : 0u);
    if constexpr (TRAIN) rw.pb = (valid ? st.db : tr) + (valid ? o : 0u);
    if constexpr (KIND == 0) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 16; j++) ss = fmaf(a0[j], a0[j], fmaf(a1[j], a1[j], ss));
      ss = g5_halfsum(ss);
      const float rinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);  // F.normalize (voicebox_pytorch.py:286); the IEEE sequence, as EpiQKV
      rw.r = fmaf(rinv, st.nscale, st.none);
      if constexpr (TRAIN) {
        const bool wr = valid && lane < 32;
        unsigned orn = (unsigned)((b * H + st.head) * Np + n);
        asm volatile("" : "+v"(orn));
        float* prn = (wr ? st.rn : reinterpret_cast<float*>(tr)) + (wr ? orn : 0u);
        *prn = rinv;
      }
    }
  }
  // LDS reads of half pr: cos / sin of the lane's token at features 16 pr + 4 hi + {0..3}, + 8
  template <int KIND>
  VBX_DEV void reads(int pr, unsigned rot_addr, f32x4 (&e)[4]) const {
    if constexpr (KIND == 0) {
      if (pr == 0) {
        G5_DS_B128(e[0], rot_addr, 0); G5_DS_B128(e[1], rot_addr ^ 32, 0); G5_DS_B128(e[2], rot_addr, 4096); G5_DS_B128(e[3], rot_addr ^ 32, 4096);
      } else {
        G5_DS_B128(e[0], rot_addr ^ 64, 0); G5_DS_B128(e[1], rot_addr ^ 96, 0); G5_DS_B128(e[2], rot_addr ^ 64, 4096); G5_DS_B128(e[3], rot_addr ^ 96, 4096);
      }
    }
  }
  template <int KIND, bool TRAIN, bool F16>
  VBX_DEV void half(const State& st, const Row& rw, int pr, const f32x16& a0, const f32x16& a1, const f32x4 (&e)[4], int lane) const {
    float olo[8], ohi[8];
    if constexpr (KIND == 1) {  // v: plain head split
#pragma unroll
      for (int i = 0; i < 8; i++) { olo[i] = a0[8 * pr + i]; ohi[i] = a1[8 * pr + i]; }
      g5_store16<G5_F16_SAT>(rw.p16, pr, olo, lane);
      g5_store16<G5_F16_SAT>(rw.p16 + 32, pr, ohi, lane);
      if constexpr (TRAIN) {
        g5_store16<G5_BF16>(rw.pb, pr, olo, lane);
        g5_store16<G5_BF16>(rw.pb + 32, pr, ohi, lane);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float lo = a0[8 * pr + i] * st.glo[8 * pr + i], hi = a1[8 * pr + i] * st.ghi[8 * pr + i];
        const float c = e[i >> 2][i & 3], s = e[2 + (i >> 2)][i & 3];
        // rotate_half (voicebox_pytorch.py:193-199), then the row's 1 / |x| * scale
        olo[i] = (lo * c - hi * s) * rw.r;
        ohi[i] = (hi * c + lo * s) * rw.r;
      }
      if constexpr (TRAIN) {
        g5_store16<G5_BF16>(rw.pb, pr, olo, lane);
        g5_store16<G5_BF16>(rw.pb + 32, pr, ohi, lane);
      }
#pragma unroll
      for (int i = 0; i < 8; i++) { olo[i] *= st.post; ohi[i] *= st.post; }
      g5_store16<G5_F16>(rw.p16, pr, olo, lane);
      g5_store16<G5_F16>(rw.p16 + 32, pr, ohi, lane);
    }
  }
};

// ---- FeedForward[0] + GEGLU.  Packed weight rows (gemm.hip EpiGEGLU): every 128 rows = 64 "x" rows then their 64 "gate" rows.
// Slab s = 32 x rows + their 32 gate rows: block 0 = x, block 1 = gate; output columns (s >> 1) * 64 + (s & 1) * 32 + 0..31.
struct Epi5GEGLU {
  u16* G; long ldg; const float* bias; u16* H1; long ldh; u16* Gb;
  static constexpr int KINDS = 1;
  static constexpr bool HAS_ROT = false;
  struct State {
    float bx[16], bg[16];
    int kind;
    int col;  // first output column of the slab
    int wx;   // first packed weight row of the x block
  };
  struct Row {
    u16 *pg, *pgb, *ph;
  };
  template <int KIND> static constexpr int nreads() { return 0; }
  VBX_DEV int wrow(int slab, int blk) const { return (slab >> 1) * 128 + blk * 64 + (slab & 1) * 32; }
  VBX_DEV void init(State& st, int slab, int lane) const {
    st.kind = 0;
    st.col = (slab >> 1) * 64 + (slab & 1) * 32;
    st.wx = wrow(slab, 0);
    g5_load16(bias + st.wx, lane, st.bx);
    g5_load16(bias + st.wx + 64, lane, st.bg);
  }
  VBX_DEV void issue_rot(char*, int, int, int, int) const {}
  template <int KIND, bool TRAIN>
  VBX_DEV void pre(const State& st, Row& rw, const f32x16&, const f32x16&, int row0, int lane, int M) const {
    const int gr = row0 + (lane & 31);
    const bool valid = gr < M;
    u16* tr = reinterpret_cast<u16*>(g5_trash) + lane * 128;
    unsigned og = (unsigned)(max(min(gr, M - 1), 0) * (int)ldg + st.col), oh = (unsigned)(max(min(gr, M - 1), 0) * (int)ldh + st.wx);  // < 2^31 (host check)
    asm volatile("" : "+v"(og), "+v"(oh));  // (computed for every lane: no exec branch inside the MFMA block)
    rw.pg = (valid ? G : tr) + (valid ? og : 0u);
    if constexpr (TRAIN) {
      rw.pgb = (valid ? Gb : tr) + (valid ? og : 0u);
      rw.ph = (valid ? H1 : tr) + (valid ? oh : 0u);
    }
  }
  template <int KIND>
  VBX_DEV void reads(int, unsigned, f32x4 (&)[4]) const {}
  template <int KIND, bool TRAIN, bool F16>
  VBX_DEV void half(const State& st, const Row& rw, int pr, const f32x16& a0, const f32x16& a1, const f32x4 (&)[4], int lane) const {
    float x[8], g[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      x[i] = a0[8 * pr + i] + st.bx[8 * pr + i];
      g[i] = a1[8 * pr + i] + st.bg[8 * pr + i];
      o[i] = gelu_erf(g[i]) * x[i];
    }
    if constexpr (F16) g5_store16<G5_F16_SAT>(rw.pg, pr, o, lane);
    else g5_store16<G5_BF16>(rw.pg, pr, o, lane);
    if constexpr (TRAIN) {
      g5_store16<G5_BF16>(rw.pgb, pr, o, lane);
      g5_store16<G5_BF16>(rw.ph, pr, x, lane);
      g5_store16<G5_BF16>(rw.ph + 64, pr, g, lane);
    }
  }
};

template <class Epi, bool F16, bool TRAIN>
__global__ __launch_bounds__(256, 1) void gemm5_kernel(G5Params p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pan = blockIdx.x / p.wpp, idx = blockIdx.x - pan * p.wpp;
  if (pan >= p.npan || idx >= p.nrb) return;
  const int nb = (p.nrb - idx + p.wpp - 1) / p.wpp;  // this workgroup's blocks: idx, idx + wpp, ...
  const int slab_raw = pan * 4 + wave;
  const bool active = slab_raw < p.nslab;
  const int slab = active ? slab_raw : 0;  // (an idle wave of the last panel repeats slab 0 into the trash)

  // ---- the stationary weight slab: 2 feature blocks x 32 k-steps of A-operand fragments (256 registers).  Loaded THROUGH the LDS:
  // a lane's fragments are 16-byte pieces of 32 different rows -- as direct global loads every instruction touches 32 cache lines
  // (13 us of prologue); as 1 KiB row DMAs into a wave-private 32 KiB region + the activation fragments' swizzled reads it is an
  // L2 -> LDS stream of 256 KiB per workgroup.
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
  if (!active) {  // idle wave: its stores go to the trash
    // (the State's output pointers are only used through Row, which pre() redirects for rows >= M: give it M = 0)
  }
  const int Meff = active ? p.M : 0;  // rows >= Meff are "invalid" for the epilogue

  auto run = [&](auto kind_c) {
    constexpr int KIND = decltype(kind_c)::value;
    constexpr int E = Epi::template nreads<KIND>();   // LDS reads of an epilogue half
    constexpr int NP = 8 + (Epi::HAS_ROT ? 2 : 0);    // LDS-DMA pieces per wave per block
    // ---- activation ring.  This wave's 8 rows of a block: t = 8 wave + q, chunk position = lane -> source chunk lane ^ (t & 15)
    const int swz0 = (wave & 1) * 8;
    auto issue = [&](int j) {  // block j of this workgroup -> X slot j % 3, rotary slot j % 4
      if constexpr (VBX_G5_ABL & 2) return;
      const int rb = idx + j * p.wpp;
      char* dst = smem + (j % G5_NSLOT) * G5_SLOT + wave * 8 * G5_ROWB;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int row = min(rb * 32 + wave * 8 + q, p.M - 1);
        const u16* src = p.A + (long)row * p.lda + ((lane ^ (swz0 + q)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + q * G5_ROWB), 16, 0, 0);
      }
      if constexpr (Epi::HAS_ROT) epi.issue_rot(smem + G5_ROT0 + (j & 3) * G5_ROTSLOT, rb * 32, wave, lane, p.M);
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
    // One phase = the MFMAs of block j (DO_MFMA) with the epilogue of block j - 1 (DO_EPI) in between, as ONE basic block.
    // LDS operations in program order (all inline asm, so the order is the source order) and the counted waits:
    //   R(xa,0) | R(xb,1) W(xa) M0 [pre] R(xa,2) W(xb) M1 | R(xb,3) E0 W(xa) M2 R(xa,4) W(xb) M3 W(E0) [half 0]
    //           | R(xb,5) E1 W(xa) M4 R(xa,6) W(xb) M5 W(E1) [half 1] | R(xb,7) W(xa) M6 W(xb) M7
    auto phase = [&](auto mf_c, auto ep_c, int j) {
      constexpr bool DO_MFMA = decltype(mf_c)::value && !(VBX_G5_ABL & 8), DO_EPI = decltype(ep_c)::value && !(VBX_G5_ABL & 1);
      constexpr int X4 = DO_MFMA ? 4 : 0, EE = DO_EPI ? E : 0;
      const unsigned so = (unsigned)((j % G5_NSLOT) * G5_SLOT);
      const unsigned ra = rot_a + (unsigned)(((j - 1) & 3) * G5_ROTSLOT);
      s16x8 xa[4], xb[4];
      f32x4 e[4];
      typename Epi::Row rw;
#define G5_READ4(x, kb)                                                          \
  if constexpr (DO_MFMA) {                                                       \
    G5_DS_B128(x[0], fa[((kb) * 4 + 0) & 7] + so, (((kb) * 4 + 0) >> 3) * 256);  \
    G5_DS_B128(x[1], fa[((kb) * 4 + 1) & 7] + so, (((kb) * 4 + 1) >> 3) * 256);  \
    G5_DS_B128(x[2], fa[((kb) * 4 + 2) & 7] + so, (((kb) * 4 + 2) >> 3) * 256);  \
    G5_DS_B128(x[3], fa[((kb) * 4 + 3) & 7] + so, (((kb) * 4 + 3) >> 3) * 256);  \
  }
#define G5_WAIT(x, n) if constexpr (DO_MFMA) g5_wait4<n>(x[0], x[1], x[2], x[3])
#define G5_MFMA4(x, kb)                                                          \
  if constexpr (DO_MFMA) {                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; i++) {                              \
      acc0 = mfma32<F16>(w[0][(kb) * 4 + i], x[i], acc0);                        \
      acc1 = mfma32<F16>(w[1][(kb) * 4 + i], x[i], acc1);                        \
    }                                                                            \
  }
      if constexpr (DO_MFMA) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc0[i] = acc1[i] = 0.f;
      }
      G5_READ4(xa, 0);
      // pair 0
      G5_READ4(xb, 1);
      G5_WAIT(xa, X4);
      G5_MFMA4(xa, 0);
      if constexpr (DO_EPI) epi.template pre<KIND, TRAIN>(st, rw, prv0, prv1, (idx + (j - 1) * p.wpp) * 32, lane, Meff);
      G5_READ4(xa, 2);
      G5_WAIT(xb, X4);
      G5_MFMA4(xb, 1);
      // pair 1 + epilogue half 0
      G5_READ4(xb, 3);
      if constexpr (EE > 0) epi.template reads<KIND>(0, ra, e);
      G5_WAIT(xa, X4 + EE);
      G5_MFMA4(xa, 2);
      G5_READ4(xa, 4);
      G5_WAIT(xb, EE + X4);
      G5_MFMA4(xb, 3);
      if constexpr (DO_EPI) {
        if constexpr (EE > 0) g5_wait4<X4>(e[0], e[1], e[2], e[3]);
        epi.template half<KIND, TRAIN, F16>(st, rw, 0, prv0, prv1, e, lane);
      }
      // pair 2 + epilogue half 1
      G5_READ4(xb, 5);
      if constexpr (EE > 0) epi.template reads<KIND>(1, ra, e);
      G5_WAIT(xa, X4 + EE);
      G5_MFMA4(xa, 4);
      G5_READ4(xa, 6);
      G5_WAIT(xb, EE + X4);
      G5_MFMA4(xb, 5);
      if constexpr (DO_EPI) {
        if constexpr (EE > 0) g5_wait4<X4>(e[0], e[1], e[2], e[3]);
        epi.template half<KIND, TRAIN, F16>(st, rw, 1, prv0, prv1, e, lane);
      }
      // pair 3
      G5_READ4(xb, 7);
      G5_WAIT(xa, X4);
      G5_MFMA4(xa, 6);
      G5_WAIT(xb, 0);
      G5_MFMA4(xb, 7);
      if constexpr (DO_MFMA && DO_EPI) {  // the epilogue's vector instructions go INTO the gaps of the MFMA stream (one wave per SIMD:
        // nothing else can fill them) instead of where the source has them
#pragma unroll
        for (int i = 0; i < 64; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, G5_VALU_PER_MFMA, 0);
        }
      }
      if constexpr (DO_MFMA) { prv0 = acc0; prv1 = acc1; }
      else if constexpr (decltype(mf_c)::value) { prv0 = acc0; prv1 = acc1; }
#undef G5_READ4
#undef G5_WAIT
#undef G5_MFMA4
    };
    // block j has landed (this wave's pieces; the barrier makes it everyone's).  At most the NP pieces of block j + 1 may stay in
    // flight: loads return in order, so <= NP outstanding operations of any kind means block j is complete -- and block j + 1 was
    // requested a whole phase ago, so in practice the allowance is what lets the previous epilogue's STORES stay in flight.
    auto top = [&](int j) {
      G5_STAMP(2 + 4 * j);
      if (j + 1 < nb) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      G5_STAMP(3 + 4 * j);
      __builtin_amdgcn_s_barrier();  // ... and every wave is done reading block j - 1 (X slot) and j - 2 (rotary slot): they take block j + 2
      G5_STAMP(4 + 4 * j);
      if (j + 2 < nb) issue(j + 2);
      G5_STAMP(5 + 4 * j);
    };
    using T = std::true_type;
    using F = std::false_type;
#pragma unroll
    for (int i = 0; i < 16; i++) acc0[i] = acc1[i] = prv0[i] = prv1[i] = 0.f;
    issue(0);
    if (nb > 1) issue(1);
    top(0);
    phase(T{}, F{}, 0);
    for (int j = 1; j < nb; j++) {
      top(j);
      phase(T{}, T{}, j);
    }
    G5_STAMP(2 + 4 * nb);
    phase(F{}, T{}, nb);
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
  if (p.npan > ncu || d->M >= (1 << 22)) return VBX_EUNSUPPORTED;
  p.wpp = ncu / p.npan;
  if (p.wpp > p.nrb) p.wpp = p.nrb;
  const int grid = p.npan * p.wpp;
  if (d->f16) return train ? launch5k<Epi, true, true>(p, epi, grid, st) : launch5k<Epi, true, false>(p, epi, grid, st);
  return train ? launch5k<Epi, false, true>(p, epi, grid, st) : launch5k<Epi, false, false>(p, epi, grid, st);
}

}  // namespace

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
  return VBX_EUNSUPPORTED;
}
