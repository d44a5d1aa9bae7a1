// bf16 MFMA GEMM for gfx950 with fused epilogues.  One kernel template, three operand layouts:
//   NT (nn.Linear forward), NN (dgrad), TN (wgrad, split-K).
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16.
// K-contiguous operands are staged [row][k] with a 16-byte XOR swizzle and read with ds_read_b128;
// K-strided operands (the transposed ones of NN/TN) are staged [k][col] and read with the gfx950
// hardware transpose read ds_read_b64_tr_b16, so no transposed copies of weights or activations
// ever exist in HBM.  The fp32 C tile is staged through LDS so every epilogue stores 16/32 B per lane.
#include "common.hpp"
#include "reduce_roles.hpp"
#include "gemm3_layout.hpp"  // g3::kc_slot / g3::kc_byte: the 128-byte-row K-contiguous layout shared with the one-round 64-deep tile
#include <stdlib.h>
#include <type_traits>

// Diagnostic build only (-DVBX_GEMM_TRACE, tools/build_trace_lib.sh): per-workgroup timestamps of gemm_kernel_v2 (entry, k-loop end,
// end; 100 MHz s_memrealtime) for tools/native/gemm_trace.cpp.
#ifdef VBX_GEMM_TRACE
static __device__ unsigned long long* g_gemm2_trace = nullptr;
extern "C" int vbx_debug_gemm2_trace(void* buf) {  // buf = [workgroups][5] u64, null to stop
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm2_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#define GEMM2_TRACE_DECL() unsigned long long gtr0 = __builtin_amdgcn_s_memrealtime(), gtr2 = 0
#define GEMM2_TRACE_MARK() gtr2 = __builtin_amdgcn_s_memrealtime()
#define GEMM2_TRACE_END()                                                                                          \
  if (g_gemm2_trace && threadIdx.x == 0) {                                                                         \
    unsigned long long* r = g_gemm2_trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;                     \
    r[0] = gtr0; r[1] = 0; r[2] = gtr2; r[3] = __builtin_amdgcn_s_memrealtime();                                   \
    r[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
  }
#else
#define GEMM2_TRACE_DECL()
#define GEMM2_TRACE_MARK()
#define GEMM2_TRACE_END()
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int CS_LD = 132;                   // floats per row of the staged C tile

// (Round 1's register-staged 128 x 128 x 64 kernel, `gemm_kernel` / VBX_GEMM_LEGACY, was removed in round 6: docs/history.md.)

struct GemmParams {
  const u16* A;
  const u16* B;
  int M, N, K;
  long lda, ldb;
  int kchunk;  // K range handled by one blockIdx.y (multiple of BK); == K when not split
  int tiles_m;
  int abl;     // timing ablations (tools only; results are wrong when != 0): 1 no DMA, 2 no LDS reads, 4 no barrier,
               // 8/16 linear A/B sources, 32 no epilogue functor, 64 no epilogue at all
  int stagger; // start-phase stagger of co-resident workgroups in 10 ns ticks (common.hpp::stagger_wait); 0 = off
};

// 16x16x32 MFMA on raw 16-bit fragments: bf16 (default) or fp16 (the q/k projection: 11-bit mantissa)
template <bool F16>
VBX_DEV f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// =====================================================================================================
// v2 main loop: LDS-DMA (global_load_lds, 16 B/lane) into a 4-stage ring of 128x128x32 k-tiles.
// Up to 3 stages (48 KiB per workgroup, 2 workgroups per CU) are in flight while a stage is consumed, which is
// what hides the ~1.5 us loaded-memory latency the 1-deep register-staged loop exposed (14-18 % MFMA utilisation
// measured).  The DMA destination is lane-linear, so bank-conflict-free layouts are obtained by permuting the
// per-lane SOURCE address and applying the same permutation on the fragment read.  Fragment reads are inline asm:
// hipcc would otherwise drain vmcnt(0) before every compiler-visible ds_read while a DMA is in flight.
// =====================================================================================================
constexpr int BK2 = 32, NST = 3;
constexpr int OP_BYTES = 8192;             // one operand stage: [128][32] or [32][128] 16-bit
constexpr int STAGE_BYTES = 2 * OP_BYTES;  // A | B
constexpr int GEMM2_LDS = NST * STAGE_BYTES;  // 48 KiB -> 3 workgroups per CU; also holds a 64-row fp32 C half tile
static_assert(GEMM2_LDS >= 64 * CS_LD * 4, "half C tile must fit");

__device__ uint4 g_zero_page[4];  // source of out-of-range lanes (K tail, ragged M/N)

VBX_DEV unsigned lds_addr(const char* p) { return (unsigned)(size_t)LDS_PTR(char, p); }

// KC stage layout: rows 2p,2p+1 share a 128-byte line; 16-byte slot (c + 4*(r&1)) ^ (p&7)
// KS stage layout: 256-byte k-rows; 32-byte pair index P ^ f(k), f(k) = (k&3) | ((k>>3)&1)<<2
VBX_DEV int ks_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// Per-thread, loop-invariant description of the LDS-DMA pieces of one operand (hoisted out of the k-loop: the 8192^3
// profile of the first version showed 3 non-MFMA VALU instructions per MFMA, almost all address arithmetic).
template <int MODE, int OUTER, int NTHR = 256>
struct DmaPlan {
  static_assert(MODE == 0 || OUTER == 128, "K-strided stages are 128 wide");
  static constexpr int N = OUTER * 4 / NTHR;  // DMA instructions per thread per stage (OUTER*4 sixteen-byte slots)
  const u16* base[N];                   // source of k-tile 0
  int kq[N];                            // MODE 0: k offset of the piece inside the tile; MODE 1: k row of the piece
  bool ok[N];                           // outer index in range
  long kstride;                         // elements per unit of k: 1 (k contiguous) or ld (k strided)
  VBX_DEV void init(const u16* __restrict__ X, long ld, int o0, int olim, int tid) {
    kstride = (MODE == 0) ? 1 : ld;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int s = i * NTHR + tid;  // 16-byte slot inside the operand stage (lane-linear: slot = base + lane)
      if (MODE == 0) {
        const int p = s >> 3, x = (s & 7) ^ (p & 7);
        const int r = 2 * p + (x >> 2);
        kq[i] = (x & 3) * 8;
        ok[i] = (o0 + r) < olim;
        base[i] = X + (long)(o0 + r) * ld + kq[i];
      } else {
        const int kr = s >> 4, pc = s & 15;
        const int P = (pc >> 1) ^ ks_f(kr);
        const int col = (2 * P + (pc & 1)) * 8;
        kq[i] = kr;
        ok[i] = (o0 + col) < olim;
        base[i] = X + (long)kr * ld + o0 + col;
      }
    }
  }
  // issue the DMAs of the k-tile starting at k0 into the operand stage at LDS byte address `dst` (wave-uniform part)
  VBX_DEV void issue(char* dst, int k0, int kend, int tid) const {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const bool in = ok[i] && (k0 + kq[i] < kend);
      const u16* src = in ? base[i] + (long)k0 * kstride : reinterpret_cast<const u16*>(g_zero_page);
      char* wave_dst = dst + (i * NTHR + (tid & ~63)) * 16;  // the DMA adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
    }
  }
};

// Fragment reads with compile-time stage/operand offsets folded into the DS immediate.
// MODE 0: the lane's address for sub-tile s is addr0 + s*1024 (8 row-pairs further, same swizzle phase).
// MODE 1: the 32-byte pair swizzle is not linear in s -> four precomputed lane addresses.
template <int MODE>
struct FragPlan {
  unsigned a[MODE == 0 ? 1 : 4];
  VBX_DEV void init(const char* smem, int woff, int lane) {
    if (MODE == 0) {
      const int r = woff + (lane & 15), c = lane >> 4, p = r >> 1;
      a[0] = lds_addr(smem + p * 128 + (((c + 4 * (r & 1)) ^ (p & 7)) << 4));
    } else {
      const int g = lane >> 4, a16 = lane & 15;
      const int kr = g * 8 + (a16 >> 2);
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int col = woff + s * 16 + 4 * (a16 & 3);
        const int ch = col >> 3;
        a[s] = lds_addr(smem + kr * 256 + ((((ch >> 1) ^ ks_f(kr)) * 2 + (ch & 1)) << 4) + (col & 7) * 2);
      }
    }
  }
  template <int OFF, int S>  // OFF: byte offset of the operand stage inside the ring
  VBX_DEV void read(bf16x8& out, s16x4& lo, s16x4& hi) const {
    if (MODE == 0) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(a[0]), "i"(OFF + S * 1024) : "memory");
    } else {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a[MODE == 0 ? 0 : S]), "i"(OFF) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a[MODE == 0 ? 0 : S]), "i"(OFF + 1024) : "memory");
    }
  }
};

// BM_ = 128: 2x2 waves of 64x64.  BM_ = 64 (MA == 0 only): 1x4 waves of 64x32 -- twice the workgroups for the
// N = dim dgrads that would otherwise fill half the chip.
// Timing ablations on 8192^3 (VBX_GEMM_ABL, tools only): full 690 TF; no LDS reads 691; no barrier 706; no DMA 1350;
// neither DMA nor LDS reads 2080 (83 % of peak) -> the L2->LDS operand stream of 128x128 tiles (64 FLOP/B) is the limiter.
template <int MA, int MB, class Epi, bool F16, int BM_>
__global__ __launch_bounds__(256, 3) void gemm_kernel_v2(GemmParams p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GEMM2_TRACE_DECL();
  if (p.stagger && gridDim.y == 1 && blockIdx.x < 768) stagger_wait(blockIdx.x >> 8, p.stagger);  // 3 workgroups per CU
#define VBX_BX_ blockIdx.x
#define VBX_T_ gridDim.x
#define VBX_SPLIT_ blockIdx.y
#include "gemm_v2_body.inc"
  GEMM2_TRACE_END();
#undef VBX_BX_
#undef VBX_T_
#undef VBX_SPLIT_
}

// ---- The one-round 160-row tile.  For the N = dim GEMMs (to_out, FeedForward-out, the dgrads into the residual width) at M = 8 x 1040 =
// 8320 rows: 128-row tiles give 65 x 4 = 260 workgroups, i.e. 4 CUs get TWO tiles and the kernel lasts as long as those; 8320 =
// 52 x 160 gives 208 equal tiles, one per CU, in one round.  (Rounds 1-2 ran it 32 deep, with 4 waves / a 5-slot ring and then 8
// waves; the 64-deep form below replaced both and they were removed in round 6 -- numbers in docs/history.md.)
// ---- 160 x 128 x 64: the one-round tile with 128-BYTE operand rows.  An LDS-DMA instruction costs the texture path one slot
// per cache line it touches: 8 rows x 128 B stream at 52 B/clk per CU from L2, the 16 rows x 64 B pieces of a 32-deep k-tile at
// 27 B/clk (tools/probes/l2_fill_rate.hip swz) -- and every k-loop of this file ran at ~18.  So the K-contiguous operands are staged
// 64 deep: LDS row = one row's 64 k values, 16-byte chunk c at c ^ (row & 7) (the layout of gemm3.hip), a stage = A [192][64]
// (24 KiB: 3 DMA instructions x 512 threads, rows >= 160 from the zero page) + B [128][64] or two K-strided [32][128] images
// (16 KiB), 3 stages = 120 KiB, one barrier per 64 k (half as many as before).  Back to back (tools/native/gemm3_check time): to_out
// 21.9 -> 18.9 us, FeedForward-out 26.3 -> 22.4, dgrad to_qkv 45.0 -> 36.4 (720 TFLOP/s), dgrad FeedForward-in 42.4 -> 34.0.
#ifndef VBX_V9_ABL
#define VBX_V9_ABL 0  // diagnostic builds (tools/native/v9_abl.sh): 1 no DMA behind the prologue, 2 no fragment reads, 4 no barrier, 8 no MFMAs, 64 no epilogue
#endif
constexpr int V9_A_BYTES = 192 * 128;
constexpr int V9_B_BYTES = 16384;
constexpr int V9_STAGE = V9_A_BYTES + V9_B_BYTES;  // 40 KiB
constexpr int GEMM_V9_LDS = 3 * V9_STAGE;          // 120 KiB
static_assert(GEMM_V9_LDS >= 160 * CS_LD * 4, "the staged C tile must fit");

template <int ROWS, int NTHR = 512>
struct DmaPlan64 {
  static constexpr int N = ROWS * 8 / NTHR;
  const u16* base[N];
  int kq[N];
  bool ok[N];
  VBX_DEV void init(const u16* __restrict__ X, long ld, int o0, int olim, int tid) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      int row, k8;
      g3::kc_slot(i * NTHR + tid, row, k8);  // slot -> (row, first k of its 8): host-checked in tests/native/gemm3_layout_check.cpp
      kq[i] = k8;
      ok[i] = (o0 + row) < olim;
      base[i] = X + (long)(o0 + row) * ld + kq[i];
    }
  }
  VBX_DEV void issue(char* dst, int k0, int kend, int tid) const {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const bool in = ok[i] && (k0 + kq[i] < kend);
      const u16* src = in ? base[i] + k0 : reinterpret_cast<const u16*>(g_zero_page);
      char* wave_dst = dst + (i * NTHR + (tid & ~63)) * 16;  // the DMA adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
    }
  }
};
struct FragPlan64 {  // 16 rows [woff + 16 S, +16) x k half KK of a [rows][64] operand stage; woff a multiple of 8
  unsigned a[2];
  VBX_DEV void init(const char* op_base, int woff, int lane) {
#pragma unroll
    for (int kk = 0; kk < 2; kk++) a[kk] = lds_addr(op_base + g3::kc_frag_byte(woff, kk, lane));
  }
  template <int S, int KK>
  VBX_DEV void read(bf16x8& out) const {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(a[KK]), "i"(S * 2048) : "memory");
  }
};

template <class Epi> struct v9_onechunk : std::false_type {};  // specialised behind the functors
template <int MB, class Epi, bool F16>
__global__ __launch_bounds__(512, 1) void gemm_kernel_bm160k64(GemmParams p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int T = gridDim.x, xcd = blockIdx.x & 7, qi = blockIdx.x >> 3;
  const int q = T >> 3, r = T & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + qi;
  const int tiles_n = T / p.tiles_m;
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * 160, n0 = tn * BN;
  const int kend = p.K;
  const int nt = (kend + 63) / 64;

  DmaPlan64<192> da;
  DmaPlan64<128> db0;            // MB == 0
  DmaPlan<1, 128, 512> db1;      // MB == 1: one 32-deep K-strided image per instruction
  da.init(p.A, p.lda, m0, min(p.M, m0 + 160), tid);
  if (MB == 0) db0.init(p.B, p.ldb, n0, p.N, tid);
  else db1.init(p.B, p.ldb, n0, p.N, tid);
  FragPlan64 fa[3], fb0[3];
  FragPlan<1> fb1[3];
#pragma unroll
  for (int s = 0; s < 3; s++) {
    fa[s].init(smem + s * V9_STAGE, wm * 80, lane);
    if (MB == 0) fb0[s].init(smem + s * V9_STAGE + V9_A_BYTES, wn * 32, lane);
    else fb1[s].init(smem + s * V9_STAGE + V9_A_BYTES, wn * 32, lane);
  }

  f32x4 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int slot, int t) {  // k-tile t -> ring slot (5 DMA instructions per thread)
    char* dst = smem + slot * V9_STAGE;
    da.issue(dst, t * 64, kend, tid);
    if (MB == 0) {
      db0.issue(dst + V9_A_BYTES, t * 64, kend, tid);
    } else {
      db1.issue(dst + V9_A_BYTES, t * 64, kend, tid);
      db1.issue(dst + V9_A_BYTES + OP_BYTES, t * 64 + 32, kend, tid);
    }
  };
  struct Frags9 { bf16x8 af[2][5], bfr[2][2]; s16x4 blo[2][2], bhi[2][2]; };
  auto read_frags = [&](auto stg_c, Frags9& f) {
    constexpr int STG = decltype(stg_c)::value;
    fa[STG].template read<0, 0>(f.af[0][0]); fa[STG].template read<1, 0>(f.af[0][1]); fa[STG].template read<2, 0>(f.af[0][2]);
    fa[STG].template read<3, 0>(f.af[0][3]); fa[STG].template read<4, 0>(f.af[0][4]);
    if (MB == 0) {
      fb0[STG].template read<0, 0>(f.bfr[0][0]); fb0[STG].template read<1, 0>(f.bfr[0][1]);
    } else {
      fb1[STG].template read<0, 0>(f.bfr[0][0], f.blo[0][0], f.bhi[0][0]); fb1[STG].template read<0, 1>(f.bfr[0][1], f.blo[0][1], f.bhi[0][1]);
    }
    fa[STG].template read<0, 1>(f.af[1][0]); fa[STG].template read<1, 1>(f.af[1][1]); fa[STG].template read<2, 1>(f.af[1][2]);
    fa[STG].template read<3, 1>(f.af[1][3]); fa[STG].template read<4, 1>(f.af[1][4]);
    if (MB == 0) {
      fb0[STG].template read<0, 1>(f.bfr[1][0]); fb0[STG].template read<1, 1>(f.bfr[1][1]);
    } else {
      fb1[STG].template read<OP_BYTES, 0>(f.bfr[1][0], f.blo[1][0], f.bhi[1][0]); fb1[STG].template read<OP_BYTES, 1>(f.bfr[1][1], f.blo[1][1], f.bhi[1][1]);
    }
  };
  auto mfmas = [&](Frags9& f) {
    if (MB == 1) {
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          s16x8 v = {f.blo[kk][s2][0], f.blo[kk][s2][1], f.blo[kk][s2][2], f.blo[kk][s2][3],
                     f.bhi[kk][s2][0], f.bhi[kk][s2][1], f.bhi[kk][s2][2], f.bhi[kk][s2][3]};
          f.bfr[kk][s2] = __builtin_bit_cast(bf16x8, v);
        }
    }
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma16<F16>(f.af[kk][i], f.bfr[kk][j], acc[i][j]);
  };
  if (nt > 0) stage(0, 0);
  if (nt > 1) stage(1, 1);
  auto step = [&](auto stg_c, int t) {
    constexpr int STG = decltype(stg_c)::value;
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // k-tile t has landed, t+1 may stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(VBX_V9_ABL & 4)) __builtin_amdgcn_s_barrier();  // k-tile t visible to all; every wave is done reading k-tile t-1
    Frags9 f;
    if ((VBX_V9_ABL & 2) && t > 0) {
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
#pragma unroll
        for (int i = 0; i < 5; i++) asm volatile("" : "=v"(f.af[kk][i]));
#pragma unroll
        for (int j = 0; j < 2; j++) asm volatile("" : "=v"(f.bfr[kk][j]), "=v"(f.blo[kk][j]), "=v"(f.bhi[kk][j]));
      }
    } else {
      read_frags(stg_c, f);
    }
    // the DMAs of k-tile t+2 are issued BEHIND the fragment reads: the LDS round trip runs under their issue time (dgrad to_qkv
    // 39.1 -> 36.4 us back to back against issuing them first).  Prefetching the fragments of k-tile t+1 before the MFMAs of
    // k-tile t (two register sets) measured 1 us SLOWER than this order.
    if (t + 2 < nt && !(VBX_V9_ABL & 1)) stage((STG + 2) % 3, t + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(VBX_V9_ABL & 8)) mfmas(f);
    else asm volatile("" ::"v"(f.af[0][0]), "v"(f.af[1][4]), "v"(f.bfr[0][0]), "v"(f.bfr[1][1]));
  };
  for (int t = 0; t < nt; t += 3) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < nt) step(std::integral_constant<int, 2>{}, t + 2);
  }
  // ---- epilogue: as gemm_kernel_bm160x8
  float* Cs = reinterpret_cast<float*>(smem);
  if (VBX_V9_ABL & 64) {
    float tsum = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) tsum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (tsum == 123.456f) Cs[tid] = tsum;
    return;
  }
  const int half = tid >> 8, tq = tid & 255;
  // 16-bit outputs: the whole 160 x 128 fp32 tile is staged at once in the idle ring (84.5 of 120 KiB), two barriers instead of six: the N = dim
  // dgrads 36.1 -> 35.6, 34.4 -> 33.6 us.  The fp32 + residual functor keeps the three 64-row chunks: staged at once its stores come in one
  // burst at the end (to_out 18.9 -> 19.3, FeedForward-out 22.4 -> 23.0 us; profiles/r06_ab_v9_onechunk.txt).
  if (v9_onechunk<Epi>::value) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
          Cs[(wm * 80 + i * 16 + (lane >> 4) * 4 + rr) * CS_LD + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][rr];
    __syncthreads();
    epi(Cs + half * 80 * CS_LD, m0 + half * 80, n0, tq, 0, p.M, p.N, 80);
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int trow = wm * 80 + i * 16;
      if ((trow >> 6) == c) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
            Cs[(trow - 64 * c + (lane >> 4) * 4 + rr) * CS_LD + wn * 32 + j * 16 + (lane & 15)] = acc[i][j][rr];
      }
    }
    __syncthreads();
    if (c < 2 || half == 0) epi(Cs + half * 32 * CS_LD, m0 + c * 64 + half * 32, n0, tq, 0, p.M, p.N, 32);
  }
}

VBX_DEV void load8(const float* Cs, int row, int cc, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc * 8);
  const float4 b = *reinterpret_cast<const float4*>(Cs + row * CS_LD + cc * 8 + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
VBX_DEV uint4 pack8_bf16(const float v[8]) {
  return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
VBX_DEV uint4 pack8_f16(const float v[8]) {
  return make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
}
VBX_DEV uint4 pack8_f16_sat(const float v[8]) {  // unbounded values: v, GEGLU output (common.hpp)
  return make_uint4(pack_f16x2_sat(v[0], v[1]), pack_f16x2_sat(v[2], v[3]), pack_f16x2_sat(v[4], v[5]), pack_f16x2_sat(v[6], v[7]));
}

// ------------------------------------------------------------------------------- epilogues
// Make the loads that produced v[] complete HERE (an empty asm that reads the registers).  The row passes below sit behind
// `if (gr < M)` guards, so hipcc cannot know whether an earlier pass already waited for these registers and would wait again in
// every pass -- with vmcnt(0), which also drains the previous pass's stores.
VBX_DEV void retire8(const float (&v)[8]) {
  asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
VBX_DEV void retire4(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
// v[0..7] = p[0..7] (p 16-byte aligned) or zeros
VBX_DEV void gload8(const float* p, bool ok, float v[8]) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (ok) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// All epilogues load what does not depend on the row (bias, qk-norm gamma) ONCE per call and request the row-dependent operands
// (residual) of several passes before the first is consumed: written naively, every pass of the row loop is a global load ->
// s_waitcnt vmcnt(0) -> use chain (hipcc does not hoist loads out of the `if (gr < M)` guard), i.e. 4-12 exposed L2 round trips
// per tile -- the GEGLU functor alone cost 14 of the FeedForward-in GEMM's 46 us.
struct EpiBF16 {
  u16* C; long ldc; const float* bias;
  VBX_DEV void operator()(const float* Cs, int m0, int n0, int tid, int, int M, int N, int rows) const {
    const int cc = tid & 15;
    const int gc = n0 + cc * 8;
    float bv[8];
    gload8(bias ? bias + gc : nullptr, bias && gc < N, bv);
    retire8(bv);
    for (int it = 0; it < rows / 16; it++) {
      const int row = it * 16 + (tid >> 4);
      const int gr = m0 + row;
      if (gr < M && gc < N) {
        float v[8];
        load8(Cs, row, cc, v);
        if (bias) {
#pragma unroll
          for (int i = 0; i < 8; i++) v[i] += bv[i];
        }
        *reinterpret_cast<uint4*>(C + (long)gr * ldc + gc) = pack8_bf16(v);
      }
    }
  }
};

#ifndef VBX_V9_ONECHUNK
#define VBX_V9_ONECHUNK 1  // 0: three chunks for every functor (A/B builds)
#endif
template <> struct v9_onechunk<EpiBF16> : std::integral_constant<bool, VBX_V9_ONECHUNK != 0> {};

struct EpiF32 {
  float* C; long ldc; const float* bias; const float* resid; u16* C2;
  VBX_DEV void operator()(const float* Cs, int m0, int n0, int tid, int, int M, int N, int rows) const {
    const int cc = tid & 15;
    const int gc = n0 + cc * 8;
    const bool colok = gc < N;
    float bv[8];
    gload8(bias ? bias + gc : nullptr, bias && colok, bv);
    const int nit = rows / 16;
    for (int it0 = 0; it0 < nit; it0 += 4) {  // up to four passes share one round of residual loads
      float4 ra[4], rb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int gr = m0 + (it0 + u) * 16 + (tid >> 4);
        if (resid && it0 + u < nit && gr < M && colok) {
          const long o = (long)gr * ldc + gc;
          ra[u] = *reinterpret_cast<const float4*>(resid + o);
          rb[u] = *reinterpret_cast<const float4*>(resid + o + 4);
        } else {
          ra[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          rb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      retire8(bv);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        retire4(ra[u]);
        retire4(rb[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int row = (it0 + u) * 16 + (tid >> 4);
        const int gr = m0 + row;
        if (it0 + u < nit && gr < M && colok) {
          float v[8];
          load8(Cs, row, cc, v);
          const long o = (long)gr * ldc + gc;
          if (bias) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] += bv[i];
          }
          if (resid) {
            v[0] += ra[u].x; v[1] += ra[u].y; v[2] += ra[u].z; v[3] += ra[u].w;
            v[4] += rb[u].x; v[5] += rb[u].y; v[6] += rb[u].z; v[7] += rb[u].w;
          }
          *reinterpret_cast<float4*>(C + o) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(C + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          if (C2) *reinterpret_cast<uint4*>(C2 + o) = pack8_bf16(v);
        }
      }
    }
  }
};

struct EpiSplitK {
  float* C;
  VBX_DEV void operator()(const float* Cs, int m0, int n0, int tid, int split, int M, int N, int rows) const {
    for (int it = 0; it < rows / 16; it++) {
      const int row = it * 16 + (tid >> 4), cc = tid & 15;
      const int gr = m0 + row, gc = n0 + cc * 8;
      if (gr < M && gc < N) {
        float v[8];
        load8(Cs, row, cc, v);
        float* o = C + ((long)split * M + gr) * N + gc;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
};

// FeedForward[0] + GEGLU (voicebox_pytorch.py:338-340,345).  Weight rows are packed so that every
// 128-column tile holds 64 "x" columns followed by their 64 "gate" columns.
struct EpiGEGLU {
  u16* G; long ldg; const float* bias; u16* H1; long ldh; u16* Gb; int g_f16;
  VBX_DEV void operator()(const float* Cs, int m0, int n0, int tid, int, int M, int N, int rows) const {
    const int cc = tid & 7;          // G part: 8 threads per row (8 "x" columns and their 8 "gate" columns each)
    const int ch = tid & 15;         // H1 part: 16 threads per row
    const int gch = n0 + ch * 8;
    float bx[8], bg[8], bh[8];       // N is a multiple of 128 here: n0 + 127 < N
    gload8(bias + n0 + cc * 8, true, bx);
    gload8(bias + n0 + 64 + cc * 8, true, bg);
    gload8(bias + gch, H1 != nullptr, bh);
    retire8(bx);
    retire8(bg);
    retire8(bh);
    for (int it = 0; it < rows / 32; it++) {
      const int row = it * 32 + (tid >> 3);
      const int gr = m0 + row;
      if (gr < M) {
        float x[8], g[8], o[8];
        load8(Cs, row, cc, x);
        load8(Cs, row, cc + 8, g);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float xv = x[i] + bx[i];
          const float gv = g[i] + bg[i];
          o[i] = gelu_erf(gv) * xv;
        }
        const long go = (long)gr * ldg + (n0 >> 1) + cc * 8;
        *reinterpret_cast<uint4*>(G + go) = g_f16 ? pack8_f16_sat(o) : pack8_bf16(o);
        if (Gb) *reinterpret_cast<uint4*>(Gb + go) = pack8_bf16(o);
      }
    }
    if (H1) {
      for (int it = 0; it < rows / 16; it++) {
        const int row = it * 16 + (tid >> 4);
        const int gr = m0 + row;
        if (gr < M) {
          float v[8];
          load8(Cs, row, ch, v);
#pragma unroll
          for (int i = 0; i < 8; i++) v[i] += bh[i];
          *reinterpret_cast<uint4*>(H1 + (long)gr * ldh + gch) = pack8_bf16(v);
        }
      }
    }
  }
};

#ifndef VBX_EPIQKV_ABL
#define VBX_EPIQKV_ABL 0  // diagnostic builds only (tools/kdim_gemm_ablation.sh): 1 no stores, 2 no 1/|x|, 4 no rotary / gamma loads
#endif
// to_qkv + MultiheadRMSNorm + rotary, written head-major (voicebox_pytorch.py:320-328).
struct EpiQKV {
  int Np, H;
  float qk_scale;
  const float* qg; const float* kg; const float* rc; const float* rs;
  u16* q16; u16* k16; u16* qb; u16* kb; u16* v; float* qrn; float* krn; u16* v16;
  float qps;  // q16 = q-hat * qps (the attention kernels' contract: scale * log2 e, include/vbx.h); qb, k16, kb unscaled
  VBX_DEV void operator()(const float* Cs, int m0, int n0, int tid, int, int M, int N, int rows) const {
    const int I = H * 64;
    const int which = n0 / I;
    const int hbase = (n0 % I) >> 6;
    // (batch, token) of a row: one division per call (m0 is uniform), rows of the chunk by a conditional subtract
    const int mb = min(m0, M - 1);
    const int b0 = mb / Np, n00 = mb - b0 * Np;
    auto split_row = [&](int gr, int& b, int& n) {
      if (rows <= Np) { b = b0; n = n00 + (gr - mb); if (n >= Np) { n -= Np; b++; } }
      else { b = gr / Np; n = gr - b * Np; }
    };
    if (which == 2) {  // v: plain head split
      for (int it = 0; it < rows / 16; it++) {
        const int row = it * 16 + (tid >> 4), cc = tid & 15;
        const int gr = m0 + row;
        if (gr >= M) continue;
        int b, n;
        split_row(gr, b, n);
        float t[8];
        load8(Cs, row, cc, t);
        const long o = (((long)b * H + hbase + (cc >> 3)) * Np + n) * 64 + (cc & 7) * 8;
        if ((VBX_EPIQKV_ABL & 1) && t[0] != 123.456f) continue;
        if (v) *reinterpret_cast<uint4*>(v + o) = pack8_bf16(t);
        if (v16) *reinterpret_cast<uint4*>(v16 + o) = pack8_f16_sat(t);
      }
      return;
    }
    // q / k: a thread owns the rotary pair of 8-wide chunks (d0..d0+7, d0+32..d0+39) of one (row, head), so rotate_half
    // needs no cross-lane traffic and the 64-wide sum of squares is a 4-lane quad reduction (DPP, no LDS permutes).
    // 8 threads per row (2 heads x 4 chunk pairs), 32 rows per pass.
    // Round 6: everything a pass reads from global memory is requested BEFORE the first pass computes -- the head's gamma chunk
    // (the same for every pass) and the rotary rows of all passes of this call.  Written pass by pass, every pass was
    // gamma loads -> s_waitcnt vmcnt(0) -> rotary loads -> s_waitcnt vmcnt(0) -> stores, and on gfx950 vmcnt also counts the previous
    // pass's STORES: two exposed round trips behind a store drain per pass, 22 of the 64 us of this GEMM (VBX_GEMM_ABL=32).
    const int hj = tid & 7;
    const int hl = hj >> 2, j = hj & 3;  // head inside the 128-column tile, chunk pair
    const int head = hbase + hl;
    const int d0 = j * 8;
    const bool isq = which == 0;
    const float* gsel = isq ? qg : kg;
    u16* dst = isq ? q16 : k16;
    u16* bcopy = isq ? qb : kb;
    float* rn = isq ? qrn : krn;
    float glo[8], ghi[8];
    gload8(gsel ? gsel + head * 64 + d0 : nullptr, qk_scale > 0.f, glo);
    gload8(gsel ? gsel + head * 64 + 32 + d0 : nullptr, qk_scale > 0.f, ghi);
    constexpr int MAXP = 2;  // passes whose loads are in flight together (the 128-wide tiles stage 64 rows per call)
    const int npass_all = rows >> 5;
    for (int it0 = 0; it0 < npass_all; it0 += MAXP) {
    const int npass = min(MAXP, npass_all - it0);
    int pb[MAXP], pn[MAXP];
    bool pv[MAXP];
    float cs[MAXP][8], sn[MAXP][8];
#pragma unroll
    for (int it = 0; it < MAXP; it++) {
      const int gr = m0 + (it0 + it) * 32 + (tid >> 3);
      pv[it] = it < npass && gr < M;
      split_row(pv[it] ? gr : mb, pb[it], pn[it]);
      gload8(rc + (long)pn[it] * 32 + d0, it < npass && !(VBX_EPIQKV_ABL & 4), cs[it]);
      gload8(rs + (long)pn[it] * 32 + d0, it < npass && !(VBX_EPIQKV_ABL & 4), sn[it]);
    }
    retire8(glo);
    retire8(ghi);
#pragma unroll
    for (int it = 0; it < MAXP; it++) {
      retire8(cs[it]);
      retire8(sn[it]);
    }
#pragma unroll
    for (int it = 0; it < MAXP; it++) {
      if (it >= npass) break;
      const int row = (it0 + it) * 32 + (tid >> 3);
      float lo[8], hi[8];
      load8(Cs, row, hl * 8 + j, lo);
      load8(Cs, row, hl * 8 + j + 4, hi);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) ss += lo[i] * lo[i] + hi[i] * hi[i];
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      // 1 / max(|x|, 1e-12) (F.normalize, voicebox_pytorch.py:286).  Tried: min(v_rsq_f32(ss), 1e12) (~20 VALU instructions fewer) --
      // the raw v_rsq moved the chaotic random-init depth-12 loss from 1.0e-3 to 3.9e-3 off the reference
      // (tests/test_model_gpu.py::test_cfg4_depth12_parity); with one Newton step the test passes again, and neither variant is
      // measurably faster (16-interval sample 80.6 vs 80.6 ms in the same run).  The IEEE sequence stays.
      const float rinv = (VBX_EPIQKV_ABL & 2) ? ss : 1.0f / fmaxf(sqrtf(ss), 1e-12f);
      if (qk_scale > 0.f) {
        const float rs_ = rinv * qk_scale;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          lo[i] = lo[i] * rs_ * glo[i];
          hi[i] = hi[i] * rs_ * ghi[i];
        }
      }
      // rotate_half (voicebox_pytorch.py:193-199): out[d] = t[d] cos - t[d+32] sin (d < 32), out[d+32] = t[d+32] cos + t[d] sin
      float olo[8], ohi[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        olo[i] = lo[i] * cs[it][i] - hi[i] * sn[it][i];
        ohi[i] = hi[i] * cs[it][i] + lo[i] * sn[it][i];
      }
      if ((VBX_EPIQKV_ABL & 1) ? (pv[it] && olo[0] == 123.456f) : pv[it]) {
        const long o = (((long)pb[it] * H + head) * Np + pn[it]) * 64 + d0;
        if (bcopy) {
          *reinterpret_cast<uint4*>(bcopy + o) = pack8_bf16(olo);
          *reinterpret_cast<uint4*>(bcopy + o + 32) = pack8_bf16(ohi);
        }
        if (isq) {
#pragma unroll
          for (int i = 0; i < 8; i++) { olo[i] *= qps; ohi[i] *= qps; }
        }
        *reinterpret_cast<uint4*>(dst + o) = pack8_f16(olo);
        *reinterpret_cast<uint4*>(dst + o + 32) = pack8_f16(ohi);
        if (rn && j == 0) rn[((long)pb[it] * H + head) * Np + pn[it]] = rinv;
      }
    }
    }
  }
};

template <int MA, int MB, bool F16 = false, class Epi>
int launch(GemmParams p, const Epi& epi, int splits, hipStream_t st) {
  static bool attr_set = false;  // >48 KiB dynamic LDS needs the opt-in once per kernel
  auto kern128 = gemm_kernel_v2<MA, MB, Epi, F16, 128>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern128), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2_LDS);
    attr_set = true;
  }
  const int tiles_n = cdiv(p.N, BN);
  // Tried and removed (numbers from the same-run A/Bs): 128x256 tiles with 4 waves of 64x128 -- main loop 30 % faster in a K
  // sweep (727 -> 935 TF/s), isolated to_qkv / FeedForward-in launches 3-8 % faster, train step 1 % SLOWER (2 instead of 3
  // workgroups per CU, the ~14 us VALU epilogues overlap less); the same tile with 8 waves of 64x64 -- back-to-back launches
  // 16 % faster (FeedForward-in 45.4 -> 38.3 us), 128-forward sample 3 % SLOWER (375 -> 387 ms); 160-row tiles with a 3-slot
  // ring for the wide GEMMs -- sample 1.5 % slower.  GEMM variants have to be judged in situ.
  if constexpr (MA == 0) {
    // one-round 160-row tiles when 128-row tiles would put two on a few CUs (see gemm_kernel_bm160).  VBX_GEMM_BM160=0/1: A/B.
    static const char* b160 = getenv("VBX_GEMM_BM160");
    const long t128 = (long)p.tiles_m * tiles_n, t160 = (long)cdiv(p.M, 160) * tiles_n;
    static const bool all160 = getenv("VBX_GEMM_BM160ALL") != nullptr;  // experiment: also the multi-round GEMMs
    // Also for half a batch (the sampler integrates the two halves concurrently, solver.py): 104 such workgroups, one per CU --
    // two of these launches from the two streams then share the chip (16-interval sample 82.7 -> 79.1 ms in the same run against
    // the 64- / 128-row tiles of gemm_kernel_v2).  VBX_BM160_MIN=<tiles>: smallest one-round grid served (257 = full batches only).
    static const long min160 = getenv("VBX_BM160_MIN") ? atol(getenv("VBX_BM160_MIN")) : 96;
    // k-loop-dominated GEMMs with a light epilogue (K >= 1024, plain bf16 / fp32 stores) also run on the 64-deep tile when they
    // need MORE than one round of 160-row tiles -- the N = 1024 GEMMs of the dim-1024 model (BASELINE config 3): dgrad
    // FeedForward-in 155 -> 132 us, FeedForward-out 96 -> 80 us, train step 20.08 -> 19.65 ms in the same run.  VBX_BM160_MULTI=0: A/B.
    static const bool multi160 = !(getenv("VBX_BM160_MULTI") && atoi(getenv("VBX_BM160_MULTI")) == 0);
    constexpr int multi_k = 1024;  // at K = 512 the same tile loses to the 128 x 256 tile (dgrad FeedForward-out 27.8 vs 20.6 us)
    const bool light = std::is_same<Epi, EpiBF16>::value || std::is_same<Epi, EpiF32>::value;
    const bool use160 = (b160 ? atoi(b160) != 0 : true) && splits == 1 &&
                        ((t128 > 256 && (t160 <= 256 || all160)) || (t160 <= 256 && t160 >= min160) ||
                         (multi160 && light && p.K >= multi_k && t160 > 256));
    if (use160) {
      static bool attr9 = false;
      auto k9 = gemm_kernel_bm160k64<MB, Epi, F16>;
      if (!attr9) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k9), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_V9_LDS);
        attr9 = true;
      }
      p.tiles_m = cdiv(p.M, 160);
      hipLaunchKernelGGL(k9, dim3(p.tiles_m * tiles_n), dim3(512), GEMM_V9_LDS, st, p, epi);
      VBX_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    bool small = false;
    if constexpr (MA == 0) {
      // N = dim GEMMs (out-proj, ff-out, dgrads into the residual width): fewer than ~1.5 workgroups per CU with
      // 128-row tiles -> halve the tile height (64x128) to fill the chip.  VBX_GEMM_BM64=0/1 forces it (A/B runs).
      static const char* force = getenv("VBX_GEMM_BM64");
      // measured (same run): NN dgrads 410 -> 500 TF, NT out-proj (K=1024) +8 %, NT ff-out (K=1408) -5 %
      small = force ? (atoi(force) != 0) : ((long)p.tiles_m * tiles_n * splits < 384 && (MB == 1 || p.K <= 1024));
      if (small) {
        static bool attr64 = false;
        auto kern64 = gemm_kernel_v2<MA, MB, Epi, F16, 64>;
        if (!attr64) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern64), hipFuncAttributeMaxDynamicSharedMemorySize, GEMM2_LDS);
          attr64 = true;
        }
        p.tiles_m = cdiv(p.M, 64);
        hipLaunchKernelGGL(kern64, dim3(p.tiles_m * tiles_n, splits), dim3(256), GEMM2_LDS, st, p, epi);
      }
    }
    // Tried (round 2, session 3): a 64-deep / two-stage / two-workgroups-per-CU form of this kernel for the NT GEMMs (128-byte
    // operand rows, the recipe that took the one-round tile from 45 to 36 us): to_qkv 52.4 -> 50.4 us back to back, but the train
    // step 9.93 -> 10.05 ms and the 16-interval sample 83.2 -> 87.9 ms in the same run -- the third co-resident workgroup (its
    // epilogue under the others' k-loops) is worth more than the cheaper staging.  Removed.
    if (!small) hipLaunchKernelGGL(kern128, dim3(p.tiles_m * tiles_n, splits), dim3(256), GEMM2_LDS, st, p, epi);
  }
  VBX_LAUNCH_CHECK();
  return 0;
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int M, int N, float* __restrict__ dst,
                                     int dst_rows, int dst_cols, long dst_ld, int rowmap, int F, int accumulate) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / N), c = (int)(i - (long)r * N);
    int dr = r;
    if (rowmap == 1) dr = geglu_row_unmap(r, F);
    if (dr < 0 || dr >= dst_rows || c >= dst_cols) continue;
    float s = 0.f;
    for (int k = 0; k < splits; k++) s += slabs[(long)k * total + i];
    float* o = dst + (long)dr * dst_ld + c;
    *o = accumulate ? (*o + s) : s;
  }
}

// several split-K reductions in one launch (the four weight gradients of a layer): job j owns blocks [block0, next block0)
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(vbx_skr_jobs jobs) {
  int j = 0;
#pragma unroll
  for (int i = 1; i < VBX_SKR_MAX; i++)
    if (i < jobs.n && (int)blockIdx.x >= jobs.job[i].block0) j = i;
  const vbx_skr_job jb = jobs.job[j];
  const float sq = skr_role(jb, ((long)(blockIdx.x - jb.block0) * 256 + threadIdx.x) * 4);  // N % 4 == 0: four columns of one row
  if (jb.sq) {  // block-uniform: the gradient-norm partial of this block (vbx_skr_job.sq)
    __shared__ float wsum[4];
    const float t = block_sum_fixed(sq, wsum);
    if (threadIdx.x == 0) jb.sq[blockIdx.x - jb.block0] = t;
  }
}

}  // namespace

static int g_gemm_path = -1;
int vbx_gemm_path() {
  if (g_gemm_path < 0) {
    const char* e = getenv("VBX_GEMM_PATH");
    const char* e3 = getenv("VBX_GEMM3");
    g_gemm_path = e ? atoi(e) : ((e3 && atoi(e3) == 0) ? 1 : 0);
    if (g_gemm_path < 0 || g_gemm_path > 4) g_gemm_path = 0;
  }
  return g_gemm_path;
}
extern "C" int vbx_gemm_select(int path) {
  VBX_REQUIRE(path >= 0 && path <= 4, "vbx_gemm_select: 0 automatic, 1 128-wide kernels only, 2 256x256 tile wherever it serves, 3 128x256 tile wherever it serves, 4 = 0 with the weight-stationary kernel forced on");
  g_gemm_path = path;
  return 0;
}
// Which tile serves a descriptor.  Automatic choice (measured in situ on the model's shapes, bench.py's stage table):
//  * the layer's four split-K weight gradients run as ONE grouped gemm3 launch (runtime.hip): 92 us against 4 x 38 us;
//  * the other NT / NN GEMMs stay on the 128-wide kernels except the two cases below: at K = dim = 512 a tile's k-loop
//    (12-14 us for 256 x 256) is followed by a VALU-bound epilogue of the same order (qk-norm + rotary 10-13 us, GEGLU 5.5 us:
//    tools/native/gemm_trace.cpp) during which the matrix pipes idle; three independent 128 x 128 workgroups per CU overlap the
//    two phases better than one 256 x 256 or two lock-stepped 128 x 256 workgroups (a start-phase stagger of the co-resident
//    workgroups, VBX_GEMM_STAGGER, did not help either); the N = dim GEMMs have too few wide tiles.
//    Paths 2 / 3 force them for measurements (tools/native/gemm3_check).
static int gemm_tile_fallback(const vbx_gemm_desc* d);
static int gemm_tile_for(const vbx_gemm_desc* d) {
  const int path = vbx_gemm_path();
  if (path == 2) return 3;
  if (path == 1) return 1;
  // K = 512 linear layers with a row-wise epilogue (to_qkv, FeedForward-in): the weight-stationary kernel (gemm5.hip).  VBX_GEMM5=0: A/B.
  static const bool g5 = !(getenv("VBX_GEMM5") && atoi(getenv("VBX_GEMM5")) == 0);
  if ((g5 || path == 4) && path != 3 && d->mode == VBX_GEMM_NT && d->K == 512 &&
      (d->epilogue == VBX_EPI_QKV || d->epilogue == VBX_EPI_GEGLU || (d->epilogue == VBX_EPI_BF16 && !d->f16 && !d->bias && d->N >= 512)))
    return 5;
  return gemm_tile_fallback(d);
}
static int gemm_tile_fallback(const vbx_gemm_desc* d) {  // the LDS-tiled kernels' choice (everything gemm5 does not serve)
  const int path = vbx_gemm_path();
  if (path == 2) return 3;
  if (path == 1) return 1;
  const bool ntnn = d->mode == VBX_GEMM_NT || d->mode == VBX_GEMM_NN;
  if (path == 3) return ntnn ? 4 : 1;
  // inference-mode FeedForward-in (GEGLU epilogue writing only the fp16 activations: no pre-activation copy, no bf16 copy) is the
  // one wide GEMM where the 128 x 256 two-per-CU tile wins: 34.1 us against 40.7 us back to back.  VBX_GEMM4_FFIN=0: A/B.
  static const bool ffin4 = !(getenv("VBX_GEMM4_FFIN") && atoi(getenv("VBX_GEMM4_FFIN")) == 0);
  if (ffin4 && d->epilogue == VBX_EPI_GEGLU && d->mode == VBX_GEMM_NT && !d->C2 && !d->C3 &&
      (long)cdiv(d->M, 128) * cdiv(d->N, 256) >= 256)
    return 4;
  // The K = dim dgrads into wide outputs (NN, plain bf16 epilogue: the to_out and FeedForward-out dgrads, N = 1024 / 1408) run on
  // the 128 x 256 tile since its epilogue stores whole rows through LDS (gemm_epi3.hpp): in the train step 24.8 -> 21.9 us and
  // 26.9 -> 25.6 us per launch, step 10.60 -> 10.48 ms in the same run.  VBX_GEMM4_DGRAD=0: A/B.  (The training FeedForward-in
  // on the same tile: 55.5 vs 55.7 us -- no change, it stays on the 128-wide kernel.)
  static const bool dgrad4 = !(getenv("VBX_GEMM4_DGRAD") && atoi(getenv("VBX_GEMM4_DGRAD")) == 0);
  if (dgrad4 && d->epilogue == VBX_EPI_BF16 && d->mode == VBX_GEMM_NN && d->K <= 512 &&
      (long)cdiv(d->M, 128) * cdiv(d->N, 256) >= 256)
    return 4;
  // (Tried: to_qkv of HALF a batch -- 792 tiles of 128 x 128 on 768 slots, 24 of them alone at the end -- as 396 tiles of 128 x 256
  //  in one round: 16-interval sample 83.2 vs 83.1 ms.  The other half batch's stream already fills that tail.)
  // (Tried: the 256 x 256 tile for the wide K = dim GEMMs of HALF a batch -- 17 x 12 / 17 x 11 tiles fit the chip in one round, and in
  //  the sampler the other half batch's stream could fill its epilogue phases: 16-interval sample 80.2 -> 80.5-82.8 ms.  No.)
  // (Tried: gemm3 for K >= 1024 with >= 256 tiles -- the dim-1024 model's to_qkv / FeedForward-in / FeedForward dgrad.  Back to
  //  back it wins (K sweep: K = 1024 68 vs 78 us); in the dim-1024 train step it lost 1.5 % in the same run, 21.1 -> 21.4 ms.)
  return 1;
}

extern "C" int vbx_gemm(const vbx_gemm_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  VBX_REQUIRE(d && d->A && d->B, "vbx_gemm: null operand");
  VBX_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "vbx_gemm: bad dims M=%d N=%d K=%d", d->M, d->N, d->K);
  VBX_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0, "vbx_gemm: leading dims must be multiples of 8 (16-byte rows)");
  VBX_REQUIRE(d->N % 8 == 0, "vbx_gemm: N must be a multiple of 8");
  int tile = gemm_tile_for(d);
  if (tile == 5) {
    const int rc = vbx_gemm5(d, st);
    if (rc != VBX_EUNSUPPORTED) return rc;
    tile = gemm_tile_fallback(d);
  }
  if (d->delta) {  // only the 128 x 256 tile's row-staged epilogue produces the attention delta (gemm_epi3.hpp): serve it there or say no
    if (tile != 4) return VBX_EUNSUPPORTED;
    return vbx_gemm4(d, st);
  }
  if (tile == 3 || tile == 4) {
    const int rc = tile == 3 ? vbx_gemm3(d, st) : vbx_gemm4(d, st);
    if (rc != VBX_EUNSUPPORTED) return rc;
  }
  GemmParams p;
  p.A = (const u16*)d->A; p.B = (const u16*)d->B;
  p.M = d->M; p.N = d->N; p.K = d->K; p.lda = d->lda; p.ldb = d->ldb;
  p.kchunk = d->K; p.tiles_m = cdiv(d->M, BM);
  static const int abl = getenv("VBX_GEMM_ABL") ? atoi(getenv("VBX_GEMM_ABL")) : 0;
  p.abl = abl;
  static const int stagger = getenv("VBX_GEMM_STAGGER") ? (int)(atof(getenv("VBX_GEMM_STAGGER")) * 100.0) : 0;
  p.stagger = stagger;
  if (d->mode == VBX_GEMM_NT) VBX_REQUIRE(d->K % 8 == 0, "vbx_gemm NT: K must be a multiple of 8");
  if (d->mode == VBX_GEMM_TN) VBX_REQUIRE(d->M % 8 == 0, "vbx_gemm TN: M must be a multiple of 8");

  switch (d->epilogue) {
    case VBX_EPI_BF16: {
      VBX_REQUIRE(d->C && d->ldc % 8 == 0, "vbx_gemm BF16: bad C/ldc");
      EpiBF16 e{(u16*)d->C, d->ldc, d->bias};
      if (d->mode == VBX_GEMM_NT) return launch<0, 0>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NN) return launch<0, 1>(p, e, 1, st);
      break;
    }
    case VBX_EPI_F32: {
      VBX_REQUIRE(d->C && d->ldc % 8 == 0, "vbx_gemm F32: bad C/ldc");
      EpiF32 e{(float*)d->C, d->ldc, d->bias, d->resid, (u16*)d->C2};
      if (d->mode == VBX_GEMM_NT && d->f16) return launch<0, 0, true>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NT) return launch<0, 0>(p, e, 1, st);
      if (d->mode == VBX_GEMM_NN) return launch<0, 1>(p, e, 1, st);
      break;
    }
    case VBX_EPI_QKV: {
      VBX_REQUIRE(d->mode == VBX_GEMM_NT, "vbx_gemm QKV: NT only");
      VBX_REQUIRE(d->H > 0 && d->H % 2 == 0 && d->N == 3 * d->H * 64, "vbx_gemm QKV: need even H and N == 3*H*64");
      VBX_REQUIRE(d->Np > 0 && d->M % d->Np == 0, "vbx_gemm QKV: M must be B*Np");
      VBX_REQUIRE(d->q16 && d->k16 && (d->v || d->v16) && d->rot_cos && d->rot_sin, "vbx_gemm QKV: null output/table");
      VBX_REQUIRE(d->qk_scale <= 0.f || (d->q_gamma && d->k_gamma), "vbx_gemm QKV: qk-norm needs gammas");
      EpiQKV e{d->Np, d->H, d->qk_scale, d->q_gamma, d->k_gamma, d->rot_cos, d->rot_sin,
               (u16*)d->q16, (u16*)d->k16, (u16*)d->qb, (u16*)d->kb, (u16*)d->v, d->q_rnorm, d->k_rnorm, (u16*)d->v16,
               d->q_prescale > 0.f ? d->q_prescale : 1.0f};
      if (d->f16) return launch<0, 0, true>(p, e, 1, st);
      return launch<0, 0>(p, e, 1, st);
    }
    case VBX_EPI_GEGLU: {
      VBX_REQUIRE(d->mode == VBX_GEMM_NT, "vbx_gemm GEGLU: NT only");
      VBX_REQUIRE(d->N % 128 == 0 && d->bias && d->C, "vbx_gemm GEGLU: N must be a multiple of 128, bias/C required");
      EpiGEGLU e{(u16*)d->C, d->ldc, d->bias, (u16*)d->C2, d->N, (u16*)d->C3, d->f16};
      if (d->f16) return launch<0, 0, true>(p, e, 1, st);
      return launch<0, 0>(p, e, 1, st);
    }
    case VBX_EPI_SPLITK: {
      VBX_REQUIRE(d->mode == VBX_GEMM_TN && d->C && d->splits >= 1, "vbx_gemm SPLITK: TN only");
      int kc = cdiv(cdiv(d->K, d->splits), BK) * BK;
      p.kchunk = kc;
      EpiSplitK e{(float*)d->C};
      return launch<1, 1>(p, e, d->splits, st);
    }
    default: break;
  }
  vbx_set_error("vbx_gemm: unsupported mode/epilogue combination (%d,%d)", d->mode, d->epilogue);
  return VBX_EUNSUPPORTED;
}

extern "C" int vbx_gemm_tn_splitk_grouped(const vbx_gemm_desc* descs, int n, void* stream) {
  VBX_REQUIRE(descs && n >= 1 && n <= 4, "vbx_gemm_tn_splitk_grouped: 1..4 jobs");
  if (vbx_gemm_path() != 1) {  // one launch of the 256 x 256 tile over all jobs (gemm3.hip)
    const int rc = vbx_gemm3_tn_splitk_grouped(descs, n, (hipStream_t)stream);
    if (rc != VBX_EUNSUPPORTED) return rc;
  }
  for (int i = 0; i < n; i++) {  // 128-wide kernels: one launch per job
    VBX_REQUIRE(descs[i].mode == VBX_GEMM_TN && descs[i].epilogue == VBX_EPI_SPLITK, "vbx_gemm_tn_splitk_grouped: job %d is not TN / SPLITK", i);
    const int rc = vbx_gemm(&descs[i], stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int vbx_splitk_reduce(const float* slabs, int splits, int M, int N, float* dst, int dst_rows, int dst_cols,
                                 int dst_ld, int rowmap, int F, int accumulate, void* stream) {
  VBX_REQUIRE(slabs && dst && splits >= 1 && M > 0 && N > 0, "vbx_splitk_reduce: bad args");
  const long total = (long)M * N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, splits, M, N, dst,
                     dst_rows, dst_cols, (long)dst_ld, rowmap, F, accumulate);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_splitk_reduce_blocks(int M, int N) { return cdiv((long)M * N / 4, 256); }
extern "C" int vbx_splitk_reduce_multi(const vbx_skr_jobs* jobs, void* stream) {
  VBX_REQUIRE(jobs && jobs->n > 0 && jobs->n <= VBX_SKR_MAX, "vbx_splitk_reduce_multi: bad job count");
  vbx_skr_jobs j = *jobs;
  int blocks = 0;
  for (int i = 0; i < j.n; i++) {
    VBX_REQUIRE(j.job[i].slabs && j.job[i].dst && j.job[i].splits >= 1 && j.job[i].M > 0 && j.job[i].N > 0 && j.job[i].N % 4 == 0,
                "vbx_splitk_reduce_multi: bad job %d (N must be a multiple of 4)", i);
    j.job[i].block0 = blocks;
    blocks += vbx_splitk_reduce_blocks(j.job[i].M, j.job[i].N);
  }
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, j);
  VBX_LAUNCH_CHECK();
  return 0;
}
