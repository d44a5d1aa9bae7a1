// gemm3: 256 x 256 x 64 MFMA GEMM tile for gfx950, 8 waves (4 x 2, each 2 x 32 rows by 128 columns), ONE workgroup per CU.
//
// Why a second GEMM kernel: the 128 x 128 / 4-wave kernels of gemm.hip are bound by the L2 -> LDS operand stream (64 FLOP per
// staged byte) and by exposed LDS round trips; all of the model's GEMMs sat at 15-21 % of the MFMA peak (profiles/r01_*).  Here
//   * the tile is 256 x 256 (128 FLOP per staged byte) and a wave owns 64 x 128 of it: 12 ds_read_b128 per 32 MFMAs;
//   * the two waves that share a SIMD (wave w and w + 4) run half a phase apart: while one issues its 16 MFMAs of a phase, the
//     other issues the next phase's fragment reads and LDS-DMAs, so the matrix pipe of every SIMD always has a wave in its
//     MFMA cluster (s_setprio 1 around the cluster lets it win the issue arbitration);
//   * a k-tile (64 deep) is 4 phases; each phase reads at most one A and one B REGION (16 KiB, gemm3_layout.hpp) and re-stages
//     one region of a later k-tile by LDS-DMA; DMA completion is awaited with a COUNTED vmcnt once per k-tile, never 0 in the
//     steady state, and barriers are raw s_barrier (an LDS-DMA in flight would turn __syncthreads() into a vmcnt(0) drain);
//   * accumulators are kept TRANSPOSED (mfma(B-fragment, A-fragment)): a lane then owns 4 CONSECUTIVE columns of one row per
//     16 x 16 block, so every epilogue (bias / residual / GEGLU / qk-norm + rotary / split-K slab) runs straight from registers
//     with 8- or 16-byte stores -- no LDS round trip, no barrier after the k-loop.
//
// Hazards of the k-loop (phase numbers global, 4 per k-tile; group G0 = waves 0-3, G1 = waves 4-7, G1 one barrier behind):
//   G0 runs load(p) in barrier interval 2p and mfma(p) in 2p+1; G1 load(p) in 2p+1 and mfma(p) in 2p+2.
//   RAW: a wave's vmcnt wait sits in ITS load(4t+3); every wave has passed it before barrier 8t+8, and the first reads of
//        k-tile t+1 are issued in load(4t+4) = interval 8t+8 (G0) / 8t+9 (G1).
//   WAR: fragment reads of phase p are complete (lgkmcnt(0) precedes the MFMAs) before barrier 2p+2 (G0) / 2p+3 (G1); the region
//        read in phase p is re-staged in phase p+2 or later, i.e. from interval 2p+4 (G0) / 2p+5 (G1) on.
//   Region read phases within k-tile t: A-lo, B-lo 4t; A-hi 4t+1; B-hi 4t+2.  Re-stage phases: A-hi(t+1) 4t, B-hi(t+1) 4t+1,
//   A-lo(t+2) 4t+2, B-lo(t+2) 4t+3.
#include "common.hpp"
#include "gemm3_layout.hpp"
#include "gemm_epi3.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

using namespace g3;
using namespace gepi;

struct G3Params {
  const u16* A;
  const u16* B;
  int M, N, K;
  long lda, ldb;
  int kchunk;   // K range of one split (multiple of BK); == K when not split
  int tiles_m;  // ceil(M / 256)
};

__device__ uint4 g3_zero_page[4];  // source of out-of-range DMA lanes
}  // namespace
#ifdef VBX_GEMM_TRACE
extern "C" int vbx_debug_gemm3_trace(void* buf) {  // diagnostic build only: buf = [workgroups][5] u64, null to stop
  return hipMemcpyToSymbol(HIP_SYMBOL(gepi::g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
namespace {

// ---------------------------------------------------------------------------------------------------------------- staging
// One operand's LDS-DMA sources.  All four pieces a thread issues per k-tile and region pair (q = 0,1 x lo,hi) derive from ONE
// source address plus wave-uniform strides, because slot s + 512 of a region is the same chunk 64 rows (KC) / 32 k rows (KS)
// further and the hi region is the lo region 128 (A) / 64 (B) outer indices further.
template <int MODE, bool IS_A>
struct Stage {
  const u16* base;  // this lane's source of the (q = 0, lo) piece of k-tile 0
  long kstep;       // elements per k-tile
  long qoff;        // q = 1 piece
  long hioff;       // hi region
  int outer0;       // global outer index of the (q = 0, lo) piece
  int kq;           // KC: k offset of the chunk inside the k-tile; KS: k row of the q = 0 piece
  int olim;
  VBX_DEV void init(const u16* __restrict__ X, long ld, int o0, int olim_, int kbeg, int tid) {
    int o, k;
    if (MODE == 0) kc_slot(tid, o, k); else ks_slot(tid, o, k);
    const int to = IS_A ? a_outer(o, 0) : b_outer(o, 0);
    outer0 = o0 + to;
    olim = olim_;
    kq = k;
    constexpr int HI = IS_A ? A_HISTEP : B_HISTEP, QS = IS_A ? A_QSTEP : B_QSTEP;
    if (MODE == 0) {
      base = X + (long)outer0 * ld + kbeg + k;
      kstep = BK;
      qoff = (long)QS * ld;  // slot + 512 -> region row + 64 -> 64 rows (A) / one wave column = 128 columns (B) further
      hioff = (long)HI * ld;
    } else {
      base = X + (long)(kbeg + k) * ld + outer0;
      kstep = (long)BK * ld;
      qoff = 32 * ld;   // slot + 512 -> k row + 32
      hioff = HI;
    }
  }
  // the two DMAs (q = 0, 1) of region `hi` of the k-tile starting at k0 (absolute), into LDS bytes [dst, dst + 16 KiB)
  template <bool FULL>
  VBX_DEV void issue(char* dst, int hi, int t, int k0, int kend, int wave) const {
    constexpr int HI = IS_A ? A_HISTEP : B_HISTEP, QS = IS_A ? A_QSTEP : B_QSTEP;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const u16* src = base + (long)t * kstep + (q ? qoff : 0) + (hi ? hioff : 0);
      if (!FULL) {
        bool ok;
        if (MODE == 0) ok = (outer0 + q * QS + hi * HI < olim) && (k0 + kq < kend);
        else ok = (outer0 + hi * HI < olim) && (k0 + kq + q * 32 < kend);
        if (!ok) src = reinterpret_cast<const u16*>(g3_zero_page);
      }
      char* wave_dst = dst + (q * THREADS + wave * 64) * 16;  // wave-uniform; the DMA adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
    }
  }
};

VBX_DEV unsigned lds_u32(const char* p) { return (unsigned)(size_t)LDS_PTR(char, p); }

// Fragment read addresses (per lane, loop invariant).  Region / fragment / k-half / buffer offsets are DS immediates except the
// buffer (64 KiB exceeds the 16-bit immediate): one address set per buffer.
template <int MODE, bool IS_A>
struct Frag {
  // KC: [kk]; KS: [fragment & (NF-1)] with NF = 2 (A: fragments of one region) or 4 (B)
  static constexpr int NA = (MODE == 0) ? 2 : (IS_A ? 2 : 4);
  unsigned a[2][NA];
  VBX_DEV void init(const char* smem, int wq, int lane) {  // wq: wave row (A, 0..3) or wave column (B, 0..1)
    const int o_w = IS_A ? wq * 32 : wq * 64;  // first region-local outer index of this wave
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int i = 0; i < NA; i++) {
        int byte;
        if (MODE == 0) byte = kc_frag_byte(o_w, i, lane);
        else byte = ks_frag_byte(o_w + i * 16, 0, lane, 0);
        a[b][i] = lds_u32(smem + b * BUF + byte);
      }
  }
};

#define G3_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define G3_DS_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")

// one MFMA fragment (raw 16-bit x 8) as it comes out of LDS
struct RawFrag {
  bf16x8 v;       // KC
  s16x4 lo, hi;   // KS
};
template <int MODE>
VBX_DEV bf16x8 frag_value(const RawFrag& f) {
  if (MODE == 0) return f.v;
  s16x8 r = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
// fragment F (index inside its region), k half KK, of the region at byte offset ROFF of buffer Bf
template <int MODE, bool IS_A, int Bf, int ROFF, int F, int KK>
VBX_DEV void read_frag(RawFrag& out, const Frag<MODE, IS_A>& fp) {
  if constexpr (MODE == 0) {
    G3_DS_B128(out.v, fp.a[Bf][KK], ROFF + F * 2048);
  } else {
    G3_DS_TR(out.lo, fp.a[Bf][F], ROFF + KK * 8192);
    G3_DS_TR(out.hi, fp.a[Bf][F], ROFF + KK * 8192 + 1024);
  }
}

// ---------------------------------------------------------------------------------------------------------------- kernel
// XCD-aware order (gemm.hip): block b runs on XCD b % 8; give every XCD one contiguous chunk of the work sequence.  Bijective.
VBX_DEV int xcd_chunk_order(int bid, int T) {
  const int xcd = bid & 7, qi = bid >> 3;
  const int q8 = T >> 3, r8 = T & 7;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + qi;
}

// one 256 x 256 output tile: `lin` = tile index (n fastest) of this GEMM, `split` = its K split
template <int MA, int MB, class Epi, bool F16>
VBX_DEV void g3_tile(const G3Params& p, const Epi& epi, char* smem, int lin, int split) {
  GEMM_TRACE_DECL();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const bool g1 = wave >= 4;  // the half of the workgroup that runs one barrier behind
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);

  Stage<MA, true> sa;
  Stage<MB, false> sb;
  sa.init(p.A, p.lda, m0, p.M, kbeg, tid);
  sb.init(p.B, p.ldb, n0, p.N, kbeg, tid);
  Frag<MA, true> fa;
  Frag<MB, false> fb;
  fa.init(smem, wr, lane);
  fb.init(smem, wc, lane);

  Acc acc;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // stage region R (0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi) of k-tile t into its buffer
  auto stage = [&](int t, int R) {
    if (t >= nt) return;  // workgroup-uniform
    char* dst = smem + (t & 1) * BUF + R * REGION;
    const int k0 = kbeg + t * BK;
    const bool full = interior && (k0 + BK <= kend);
    if (R < 2) {
      if (full) sa.template issue<true>(dst, R & 1, t, k0, kend, wave);
      else sa.template issue<false>(dst, R & 1, t, k0, kend, wave);
    } else {
      if (full) sb.template issue<true>(dst, R & 1, t, k0, kend, wave);
      else sb.template issue<false>(dst, R & 1, t, k0, kend, wave);
    }
  };

  // ---- prologue: k-tile 0 whole, A-lo / B-lo of k-tile 1
  stage(0, 0); stage(0, 2); stage(0, 1); stage(0, 3);
  stage(1, 0); stage(1, 2);
  // k-tile 0 must have landed; A-lo / B-lo of k-tile 1 stay in flight
  if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  GEMM_TRACE_MARK(gtr1);
  __builtin_amdgcn_sched_barrier(0);
  if (g1) __builtin_amdgcn_s_barrier();  // stagger: G1's phases start one barrier interval after G0's
  __builtin_amdgcn_sched_barrier(0);

  RawFrag alo[2][2], ahi[2][2], bq[4][2];  // [fragment][k half]

#define G3_BAR()                          \
  do {                                    \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);    \
  } while (0)
#define G3_MFMA_BEGIN()                                      \
  do {                                                       \
    G3_BAR();                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);                       \
    __builtin_amdgcn_s_setprio(1);                           \
  } while (0)
// ONE DMA wait per k-tile (phase 3): k-tile t+1 complete, A-lo / B-lo of t+2 (4 instructions) stay in flight.  Tried: a counted
// wait in every phase with four regions (64 KiB instead of 32 KiB) in flight -- 8192^3 1293 -> 1240 TFLOP/s, grouped wgrads 89 ->
// 94 us: the k-loop is not bound by load latency but by the L2 -> LDS fill rate (~10 TB/s aggregate: 655 / 870 / 1300 TFLOP/s for
// the 128^2, 128 x 256 and 256^2 tiles are all 10.2 TB/s times their FLOP per staged byte).
// Tried (round 2, session 3): issuing a phase's two LDS-DMAs INSIDE its MFMA cluster (after the 4th and 10th MFMA; k-tile wait
// vmcnt(2)) instead of behind the fragment reads of the load part, on the theory that the load group's 12 ds_read + 2 DMA issue
// (~440 cycles per barrier interval against 256 cycles of MFMAs) bounds the interval: correct, and SLOWER -- 8192^3 1315 -> 1018
// TFLOP/s, K = 512 at the to_qkv shape 36.8 -> 46.2 us; one piece in each part: 1183 TFLOP/s, 42.5 us.  A DMA issued by the wave
// that holds the matrix pipe stalls that pipe; the load part is the right place.
#define G3_MFMA_END()                     \
  do {                                    \
    __builtin_amdgcn_s_setprio(0);        \
    G3_BAR();                             \
  } while (0)

  // one k-tile in buffer Bf (compile time: every LDS offset is an immediate)
  auto ktile = [&](auto bf_c, int t) {
    constexpr int Bf = decltype(bf_c)::value;
    // ---- phase 0: B-lo, A-lo -> acc[0..1][0..3]
    read_frag<MB, false, Bf, OFF_BLO, 0, 0>(bq[0][0], fb); read_frag<MB, false, Bf, OFF_BLO, 0, 1>(bq[0][1], fb);
    read_frag<MB, false, Bf, OFF_BLO, 1, 0>(bq[1][0], fb); read_frag<MB, false, Bf, OFF_BLO, 1, 1>(bq[1][1], fb);
    read_frag<MB, false, Bf, OFF_BLO, 2, 0>(bq[2][0], fb); read_frag<MB, false, Bf, OFF_BLO, 2, 1>(bq[2][1], fb);
    read_frag<MB, false, Bf, OFF_BLO, 3, 0>(bq[3][0], fb); read_frag<MB, false, Bf, OFF_BLO, 3, 1>(bq[3][1], fb);
    read_frag<MA, true, Bf, OFF_ALO, 0, 0>(alo[0][0], fa); read_frag<MA, true, Bf, OFF_ALO, 0, 1>(alo[0][1], fa);
    read_frag<MA, true, Bf, OFF_ALO, 1, 0>(alo[1][0], fa); read_frag<MA, true, Bf, OFF_ALO, 1, 1>(alo[1][1], fa);
    stage(t + 1, 1);
    G3_MFMA_BEGIN();
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = mfma16<F16>(frag_value<MB>(bq[j][kk]), frag_value<MA>(alo[i][kk]), acc[i][j]);
    G3_MFMA_END();
    // ---- phase 1: A-hi -> acc[2..3][0..3]
    read_frag<MA, true, Bf, OFF_AHI, 0, 0>(ahi[0][0], fa); read_frag<MA, true, Bf, OFF_AHI, 0, 1>(ahi[0][1], fa);
    read_frag<MA, true, Bf, OFF_AHI, 1, 0>(ahi[1][0], fa); read_frag<MA, true, Bf, OFF_AHI, 1, 1>(ahi[1][1], fa);
    stage(t + 1, 3);
    G3_MFMA_BEGIN();
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[2 + i][j] = mfma16<F16>(frag_value<MB>(bq[j][kk]), frag_value<MA>(ahi[i][kk]), acc[2 + i][j]);
    G3_MFMA_END();
    // ---- phase 2: B-hi (replaces B-lo in registers) -> acc[2..3][4..7]
    read_frag<MB, false, Bf, OFF_BHI, 0, 0>(bq[0][0], fb); read_frag<MB, false, Bf, OFF_BHI, 0, 1>(bq[0][1], fb);
    read_frag<MB, false, Bf, OFF_BHI, 1, 0>(bq[1][0], fb); read_frag<MB, false, Bf, OFF_BHI, 1, 1>(bq[1][1], fb);
    read_frag<MB, false, Bf, OFF_BHI, 2, 0>(bq[2][0], fb); read_frag<MB, false, Bf, OFF_BHI, 2, 1>(bq[2][1], fb);
    read_frag<MB, false, Bf, OFF_BHI, 3, 0>(bq[3][0], fb); read_frag<MB, false, Bf, OFF_BHI, 3, 1>(bq[3][1], fb);
    stage(t + 2, 0);
    G3_MFMA_BEGIN();
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[2 + i][4 + j] = mfma16<F16>(frag_value<MB>(bq[j][kk]), frag_value<MA>(ahi[i][kk]), acc[2 + i][4 + j]);
    G3_MFMA_END();
    // ---- phase 3: no reads -> acc[0..1][4..7]; wait for k-tile t+1 (A-lo / B-lo of t+2 may stay in flight)
    stage(t + 2, 2);
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G3_MFMA_BEGIN();
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][4 + j] = mfma16<F16>(frag_value<MB>(bq[j][kk]), frag_value<MA>(alo[i][kk]), acc[i][4 + j]);
    G3_MFMA_END();
  };

  for (int t = 0; t < nt; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) ktile(std::integral_constant<int, 1>{}, t + 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (!g1) __builtin_amdgcn_s_barrier();  // pairs with G1's extra barrier: every wave has executed the same number
  __builtin_amdgcn_sched_barrier(0);

  GEMM_TRACE_MARK(gtr2);
  epi(acc, m0 + wr * 32, n0 + wc * 128, lane, split, p.M, p.N, 128, lds_u32(smem + wave * EPI_STAGE_BYTES));
  GEMM_TRACE_END();
}

template <int MA, int MB, class Epi, bool F16>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(G3Params p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  g3_tile<MA, MB, Epi, F16>(p, epi, smem, xcd_chunk_order(blockIdx.x, gridDim.x), blockIdx.y);
}

// Several GEMMs of one kind in ONE grid: job j owns work items [item0[j], item0[j+1]), an item is (split, tile) with the tile
// fastest.  First use: the four weight-gradient GEMMs of a layer (66 tiles of 256 x 256 in all, K = tokens): one launch with
// 3 K-splits each fills 198 CUs for 44 k-tiles instead of four launches of 24 / 22 / 8 / 12 tiles.
constexpr int G3_MAX_JOBS = 4;
template <class Epi>
struct G3Group {
  int n;
  int item0[G3_MAX_JOBS + 1];
  int tiles[G3_MAX_JOBS];
  G3Params p[G3_MAX_JOBS];
  Epi epi[G3_MAX_JOBS];
};
template <int MA, int MB, class Epi, bool F16>
__global__ __launch_bounds__(512, 2) void gemm3_grouped_kernel(G3Group<Epi> g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int item = xcd_chunk_order(blockIdx.x, gridDim.x);
  int j = 0;
#pragma unroll
  for (int i = 1; i < G3_MAX_JOBS; i++)
    if (i < g.n && item >= g.item0[i]) j = i;
  const int local = item - g.item0[j];
  const int split = local / g.tiles[j];
  g3_tile<MA, MB, Epi, F16>(g.p[j], g.epi[j], smem, local - split * g.tiles[j], split);
}

template <int MA, int MB, bool F16 = false, class Epi>
int launch3(G3Params p, const Epi& epi, int splits, hipStream_t st) {
  auto kern = gemm3_kernel<MA, MB, Epi, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vbx_set_error("gemm3: cannot opt in to %d bytes of LDS: %s", LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  const int tiles_n = cdiv(p.N, BN);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * tiles_n, splits), dim3(THREADS), LDS_BYTES, st, p, epi);
  VBX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// n (1..4) TN / VBX_EPI_SPLITK GEMMs in one launch of the 256 x 256 tile (slab layout and results as n vbx_gemm calls)
int vbx_gemm3_tn_splitk_grouped(const vbx_gemm_desc* descs, int n, hipStream_t st) {
  if (!descs || n < 1 || n > G3_MAX_JOBS) return VBX_EUNSUPPORTED;
  G3Group<Epi3SplitK> g;
  g.n = n;
  int items = 0;
  for (int i = 0; i < G3_MAX_JOBS; i++) {
    const vbx_gemm_desc* d = descs + (i < n ? i : 0);
    if (i < n) {
      if (d->mode != VBX_GEMM_TN || d->epilogue != VBX_EPI_SPLITK || !d->A || !d->B || !d->C || d->splits < 1) return VBX_EUNSUPPORTED;
      if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->lda % 8 || d->ldb % 8 || d->N % 8 || d->M % 8) return VBX_EUNSUPPORTED;
    }
    G3Params& p = g.p[i];
    p.A = (const u16*)d->A; p.B = (const u16*)d->B;
    p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb;
    p.kchunk = cdiv(cdiv(d->K, d->splits), BK) * BK;
    p.tiles_m = cdiv(d->M, BM);
    g.epi[i] = Epi3SplitK{(float*)d->C};
    g.tiles[i] = p.tiles_m * cdiv(d->N, BN);
    g.item0[i] = items;
    if (i < n) items += g.tiles[i] * d->splits;
  }
  g.item0[G3_MAX_JOBS] = items;
  auto kern = gemm3_grouped_kernel<1, 1, Epi3SplitK, false>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vbx_set_error("gemm3 grouped: cannot opt in to %d bytes of LDS: %s", LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(items), dim3(THREADS), LDS_BYTES, st, g);
  VBX_LAUNCH_CHECK();
  return 0;
}

// Same contract as vbx_gemm (include/vbx.h); returns VBX_EUNSUPPORTED for descriptors this tile does not serve so that the caller
// can fall back to the 128-wide kernels of gemm.hip.
int vbx_gemm3(const vbx_gemm_desc* d, hipStream_t st) {
  if (!d || !d->A || !d->B || d->M <= 0 || d->N <= 0 || d->K <= 0) return VBX_EUNSUPPORTED;
  if (d->lda % 8 || d->ldb % 8 || d->N % 8 || d->K % 8) return VBX_EUNSUPPORTED;
  if (d->mode == VBX_GEMM_TN && d->M % 8) return VBX_EUNSUPPORTED;
  G3Params p;
  p.A = (const u16*)d->A; p.B = (const u16*)d->B;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb;
  p.kchunk = d->K; p.tiles_m = cdiv(d->M, BM);
  switch (d->epilogue) {
    case VBX_EPI_BF16: {
      if (!d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
      Epi3BF16 e{(u16*)d->C, d->ldc, d->bias};
      if (d->mode == VBX_GEMM_NT && !d->f16) return launch3<0, 0>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NN) return launch3<0, 1>(p, e, 1, st);
      break;
    }
    case VBX_EPI_F32: {
      if (!d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
      Epi3F32 e{(float*)d->C, d->ldc, d->bias, d->resid, (u16*)d->C2};
      if (d->mode == VBX_GEMM_NT && d->f16) return launch3<0, 0, true>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NT) return launch3<0, 0>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NN) return launch3<0, 1>(p, e, 1, st);
      break;
    }
    case VBX_EPI_QKV: {
      if (d->mode != VBX_GEMM_NT || d->H <= 0 || d->H % 2 || d->N != 3 * d->H * 64 || d->Np <= 0 || d->M % d->Np) return VBX_EUNSUPPORTED;
      if (!d->q16 || !d->k16 || !(d->v || d->v16) || !d->rot_cos || !d->rot_sin) return VBX_EUNSUPPORTED;
      if (d->qk_scale > 0.f && !(d->q_gamma && d->k_gamma)) return VBX_EUNSUPPORTED;
      Epi3QKV e{d->Np, d->H, d->qk_scale, d->q_gamma, d->k_gamma, d->rot_cos, d->rot_sin,
                (u16*)d->q16, (u16*)d->k16, (u16*)d->qb, (u16*)d->kb, (u16*)d->v, d->q_rnorm, d->k_rnorm, (u16*)d->v16,
                d->q_prescale > 0.f ? d->q_prescale : 1.0f};
      if (d->f16) return launch3<0, 0, true>(p, e, 1, st);
      return launch3<0, 0>(p, e, 1, st);
    }
    case VBX_EPI_GEGLU: {
      if (d->mode != VBX_GEMM_NT || d->N % 128 || !d->bias || !d->C) return VBX_EUNSUPPORTED;
      Epi3GEGLU e{(u16*)d->C, d->ldc, d->bias, (u16*)d->C2, d->N, (u16*)d->C3, d->f16};
      if (d->f16) return launch3<0, 0, true>(p, e, 1, st);
      return launch3<0, 0>(p, e, 1, st);
    }
    case VBX_EPI_SPLITK: {
      if (d->mode != VBX_GEMM_TN || !d->C || d->splits < 1) return VBX_EUNSUPPORTED;
      p.kchunk = cdiv(cdiv(d->K, d->splits), BK) * BK;
      Epi3SplitK e{(float*)d->C};
      return launch3<1, 1>(p, e, d->splits, st);
    }
    default: break;
  }
  return VBX_EUNSUPPORTED;
}
