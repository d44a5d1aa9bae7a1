// Fused attention for gfx950: Attend.forward math path (attend.py:121-135) without materialising
// the (Np x Np) score matrix, plus its backward (two kernels: dq, and dk/dv).
//
// Everything is "transposed" so that the softmax axis is lane-local:
//   S^T[key][q] = K . Q^T   (v_mfma_f32_32x32x16_f16; q-hat/k-hat are fp16: logits are 10 * q.k with
//                            |q|=|k|=8, std ~80 -- bf16 operands would perturb them by O(0.1))
//   O^T[d][q]   = V^T . P^T (v_mfma_f32_32x32x16_bf16; V^T fragments come straight from the row-major
//                            V tile through ds_read_b64_tr_b16)
// With the 32x32 C layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) a lane owns ONE
// query column of S^T and of O^T: row max / sum / rescale are per-lane scalars plus one lane^32 exchange,
// and the P^T accumulator registers are already the B operand of the next MFMA (slot s of half hi <->
// key (s&3) + 8*(s>>2) + 4*hi inside each 16-key group; the V^T fragment uses the same key order).
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TILE16 = 64 * 128;  // one [64 rows][64 x 16-bit] tile, 16-byte XOR swizzle: 8192 B
constexpr float NEG_INF = -__builtin_inff();

// 16-byte-chunk swizzle of the [64][128 B] K/V/Q/dO tiles.  Two 128-byte rows span the 64 LDS banks, so same-parity rows
// compete for the same 32 banks and the XOR key must separate them for BOTH access shapes used on these tiles:
//  * ds_read_b128 row fragments: a 16-lane service group holds 8 same-parity rows with p = row>>1 covering all of p mod 8
//    -> the key must be a bijection of p mod 8;
//  * ds_read_b64_tr_b16 transposed fragments: a 32-lane group holds rows r, r+2 (p, p+1) reading the same 4 chunks
//    -> key(p) ^ key(p+1) must flip bit 2 (move the other row to the other 64-byte half).
// key = (p&1)<<2 | (p>>1)&3 does both (the previous key, row&7, was 2-way conflicted on every K and V^T read;
// SQ_LDS_BANK_CONFLICT was 2.5 cycles per LDS instruction).  key(row+16) == key(row), key(row+8) != key(row).
VBX_DEV int attn_swz(int row) {
  const int p = row >> 1;
  return ((p & 1) << 2) | ((p >> 1) & 3);
}
VBX_DEV int swz_off2(int row, int chunk) { return row * 128 + ((chunk ^ attn_swz(row)) << 4); }
VBX_DEV bf16x8 pack_frag(const f32x16& p, int t2) {
  bf16x8 r;
#pragma unroll
  for (int s = 0; s < 8; s++) r[s] = (__bf16)p[8 * t2 + s];
  return r;
}

VBX_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// P^T (or dS^T) accumulator registers 8*t2 .. 8*t2+7 -> fp16 MFMA operand; v_cvt_pkrtz packs two conversions per
// instruction (round toward zero: a 2^-12 relative bias on softmax weights that sum to 1 -- far below the fp16 noise)
VBX_DEV f16x8 pack_frag_f16_fast(const f32x16& p, int t2) {
  typedef __attribute__((ext_vector_type(2))) __fp16 h2;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const h2 a = __builtin_amdgcn_cvt_pkrtz(p[8 * t2 + 0], p[8 * t2 + 1]);
  const h2 b = __builtin_amdgcn_cvt_pkrtz(p[8 * t2 + 2], p[8 * t2 + 3]);
  const h2 c = __builtin_amdgcn_cvt_pkrtz(p[8 * t2 + 4], p[8 * t2 + 5]);
  const h2 d = __builtin_amdgcn_cvt_pkrtz(p[8 * t2 + 6], p[8 * t2 + 7]);
  const u32x4 w = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c),
                   __builtin_bit_cast(unsigned, d)};
  return __builtin_bit_cast(f16x8, w);
}

VBX_DEV int acc_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

// Epilogue helper for the backward kernels: a wave's two f32x16 accumulators hold X^T[d][row] (lane = row, 4 consecutive d
// per register group).  Stage them as row-major [32 rows][64 fp32] in the wave's private LDS area (16-byte chunks XOR-
// swizzled by row&15) and write each 256-byte row with coalesced 16-byte stores: 8 stores per lane instead of 16 scattered.
VBX_DEV void store_rows_f32(char* wst, const f32x16 (&acc)[2], float scale, float* __restrict__ gbase, int row0, int row_lim,
                            int lane) {
  const int rl = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int db = 0; db < 2; db++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int c = (db * 32 + 8 * g4 + 4 * hi) >> 2;
      *reinterpret_cast<float4*>(wst + rl * 256 + ((c ^ (rl & 15)) << 4)) =
          make_float4(acc[db][4 * g4] * scale, acc[db][4 * g4 + 1] * scale, acc[db][4 * g4 + 2] * scale, acc[db][4 * g4 + 3] * scale);
    }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int row = it * 4 + (lane >> 4), ch = lane & 15;
    if (row0 + row < row_lim)
      *reinterpret_cast<float4*>(gbase + (long)(row0 + row) * 64 + ch * 4) =
          *reinterpret_cast<const float4*>(wst + row * 256 + ((ch ^ (row & 15)) << 4));
  }
}

// Fused epilogue of the backward kernels (replaces store_rows_f32 + the separate vbx_qknorm_rope_bwd pass, i.e. a 68 MB fp32
// write and re-read of dq|dk per layer): the wave's dX^T block is staged row-major in LDS as before, then every row goes through
// the backward of rotary + MultiheadRMSNorm (voicebox_pytorch.py:193-199, 280-287, 320-328) and leaves as bf16 straight into the
// d(qkv) operand of the projection's dgrad / wgrad.  8 lanes per row, a lane owns 8 dims d0..d0+7 and reads its rotary partner
// chunk (d0 ^ 32) from LDS / global itself -- no cross-lane traffic except the 64-wide dot (quad-style xor 1,2,4).
// The gamma gradient is reduced over the workgroup's rows and written as one partial record per (which, batch, tile, head).
struct QKBwd {
  const u16* xh;        // normalised + rotated q^ or k^ (fp16) [B,H,Np,64]
  const float* rn;      // 1/|x| per row [B,H,Np] (qk-norm only)
  const float* gam;     // gamma [H,64] (qk-norm only)
  const float* rc;      // rotary cos / sin tables [Np,32]
  const float* rs;
  float qk_scale;       // 8 with qk-norm, 0 without
  u16* dqkv;            // bf16 [B*Np, ld]
  int ld, col0;         // column offset of this tensor's block inside a dqkv row (0 for q, H*64 for k)
  float* gpart;         // [B][tiles][H][64] partial gamma gradients of this tensor (qk-norm only)
  float xh_inv;         // 1 / (the factor xh carries): 1 / (scale * log2 e) for the pre-scaled q16, 1 for k16
};
// `staged`: the wave's rows already sit in wst in the layout below, scaled (the ragged tail roles of round 6 sum four waves' partial
// blocks there); acc / scale are then ignored.
VBX_DEV void store_rows_qknorm(char* wst, float* red /* [4][64] workgroup scratch */, const f32x16 (&acc)[2], float scale,
                               const QKBwd& f, bool active, int b, int h, int H, int tile, int ntiles, int row0, int Np, int lane,
                               int wave, bool staged = false) {
  const long bh = (long)b * H + h;
  float gacc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) gacc[i] = 0.f;
  if (active) {
    const int rl = lane & 31, hi = lane >> 5;
    if (!staged) {
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int c = (db * 32 + 8 * g4 + 4 * hi) >> 2;
        *reinterpret_cast<float4*>(wst + rl * 256 + ((c ^ (rl & 15)) << 4)) =
            make_float4(acc[db][4 * g4] * scale, acc[db][4 * g4 + 1] * scale, acc[db][4 * g4 + 2] * scale, acc[db][4 * g4 + 3] * scale);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int sub = lane & 7, d0 = sub * 8, dp0 = d0 ^ 32;
    const float sgn = d0 < 32 ? 1.f : -1.f;  // transpose of rotate_half
    float gm[8];
#pragma unroll
    for (int i = 0; i < 8; i++) gm[i] = (f.qk_scale > 0.f) ? f.gam[h * 64 + d0 + i] : 1.f;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = it * 8 + (lane >> 3);
      const int n = row0 + row;
      const bool valid = n < Np;
      const int nc = valid ? n : (Np - 1);
      float g[8], gp[8], qh[8], qp[8];
      {
        const int c0 = d0 >> 2, c1 = dp0 >> 2;
        const float4 a0 = *reinterpret_cast<const float4*>(wst + row * 256 + ((c0 ^ (row & 15)) << 4));
        const float4 a1 = *reinterpret_cast<const float4*>(wst + row * 256 + (((c0 + 1) ^ (row & 15)) << 4));
        const float4 p0 = *reinterpret_cast<const float4*>(wst + row * 256 + ((c1 ^ (row & 15)) << 4));
        const float4 p1 = *reinterpret_cast<const float4*>(wst + row * 256 + (((c1 + 1) ^ (row & 15)) << 4));
        g[0] = a0.x; g[1] = a0.y; g[2] = a0.z; g[3] = a0.w; g[4] = a1.x; g[5] = a1.y; g[6] = a1.z; g[7] = a1.w;
        gp[0] = p0.x; gp[1] = p0.y; gp[2] = p0.z; gp[3] = p0.w; gp[4] = p1.x; gp[5] = p1.y; gp[6] = p1.z; gp[7] = p1.w;
        const long ro = (bh * Np + nc) * 64;
        const uint4 hq = *reinterpret_cast<const uint4*>(f.xh + ro + d0);
        const uint4 hp = *reinterpret_cast<const uint4*>(f.xh + ro + dp0);
        const unsigned w[4] = {hq.x, hq.y, hq.z, hq.w}, wp[4] = {hp.x, hp.y, hp.z, hp.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          qh[2 * i] = f16_to_f32((u16)(w[i] & 0xffff));
          qh[2 * i + 1] = f16_to_f32((u16)(w[i] >> 16));
          qp[2 * i] = f16_to_f32((u16)(wp[i] & 0xffff));
          qp[2 * i + 1] = f16_to_f32((u16)(wp[i] >> 16));
        }
      }
      const float* cp = f.rc + (long)nc * 32 + (d0 & 31);
      const float* sp = f.rs + (long)nc * 32 + (d0 & 31);
      float dy[8], yv[8], out[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        dy[i] = g[i] * cp[i] + sgn * gp[i] * sp[i];
        yv[i] = (qh[i] * cp[i] + sgn * qp[i] * sp[i]) * f.xh_inv;
      }
      if (f.qk_scale > 0.f) {
        const float rinv = f.rn[bh * Np + nc];
        float u[8], du[8], dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float sg = f.qk_scale * gm[i];
          u[i] = (fabsf(sg) > 1e-20f) ? yv[i] / sg : 0.f;
          if (valid) gacc[i] += dy[i] * u[i] * f.qk_scale;
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
        u16* o = f.dqkv + ((long)b * Np + n) * f.ld + f.col0 + h * 64 + d0;
        *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf16x2(out[0], out[1]), pack_bf16x2(out[2], out[3]),
                                                  pack_bf16x2(out[4], out[5]), pack_bf16x2(out[6], out[7]));
      }
    }
  }
  if (f.qk_scale > 0.f && f.gpart) {  // gamma gradient: 8 row slots per wave -> lanes 0..7, then the four waves
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float v = gacc[i];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      gacc[i] = v;
    }
    __syncthreads();  // every wave is past its LDS staging reads
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < 8; i++) red[wave * 64 + lane * 8 + i] = gacc[i];
    }
    __syncthreads();
    if (threadIdx.x < 64)
      f.gpart[(((long)b * ntiles + tile) * H + h) * 64 + threadIdx.x] =
          red[threadIdx.x] + red[64 + threadIdx.x] + red[128 + threadIdx.x] + red[192 + threadIdx.x];
  }
}

// Workgroup -> (128-row tile, head, batch).  Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with a private
// L2; the tiles of one (batch, head) all read the same K/V (or Q/dO) panels, so they are given ids that land on ONE XCD
// and run back to back there.  With the plain (tile, h, b) grid the 9 tiles of a head sat on 9 different XCDs and every
// panel was fetched 8-9x from HBM/MALL (rocprofv3 FETCH_SIZE: 290 MB per forward launch against 51 MB of q|k|v).
struct AttnCoord { int tile, h, b; bool ok; };
// blockIdx -> (128-row tile, head, batch).  The ragged tail tile of every (b, h) (Np % 128 rows: the 16 register-token rows at the
// benchmark's Np = 1040) is a workgroup with one active wave that still walks the whole key loop: it lives about as long as a full workgroup
// (it is bound by the tile round trip, not by throughput) while using a fraction of a CU.  Such tails go FIRST, where they
// overlap with full workgroups; dispatched last they were a 30 us drain phase with the chip empty (tools/attn_timeline.py).  The short
// <= 16-row tails of round 6 go LAST instead (attn_xmap_for).
// xmap bit 0: consecutive ids rotate over the 8 XCDs, all tiles of a (b, h) on one XCD sharing its L2 copy of K / V;
// bit 1: tails last.
VBX_DEV int attn_tail_ids(int Np, int BH, int xmap) { return (Np & 127) ? ((xmap & 1) ? ((BH + 7) >> 3) * 8 : BH) : 0; }
VBX_DEV AttnCoord attn_coord_id(int id, int H, int Np, int BH, int xmap) {
  const int nfull = Np >> 7;
  const int per_tile = (xmap & 1) ? ((BH + 7) >> 3) * 8 : BH;
  const int ntail = (Np & 127) ? per_tile : 0;
  AttnCoord c;
  int bh;
  bool tail;
  if (xmap & 2) {
    tail = id >= per_tile * nfull;
    id -= tail ? per_tile * nfull : 0;
  } else {
    tail = id < ntail;
    id -= tail ? 0 : ntail;
  }
  if (tail) {
    bh = id;  // == (id >> 3) * 8 + (id & 7)
    c.tile = nfull;
  } else if (xmap & 1) {
    const int slot = id >> 3;
    bh = (slot / nfull) * 8 + (id & 7);
    c.tile = slot % nfull;
  } else {
    bh = id / nfull;
    c.tile = id - bh * nfull;
  }
  c.ok = bh < BH;
  c.b = bh / H;
  c.h = bh - c.b * H;
  return c;
}
VBX_DEV AttnCoord attn_coord(int H, int Np, int BH, int xmap) { return attn_coord_id(blockIdx.x, H, Np, BH, xmap); }


// Diagnostic build only (-DVBX_ATTN_TRACE, tools/build_trace_lib.sh): every workgroup records when it started, when its main loop
// ended and when it finished (100 MHz s_memrealtime ticks) plus where it ran, for tools/attn_timeline.py.
#ifdef VBX_ATTN_TRACE
__device__ unsigned long long* g_attn_trace = nullptr;
#define ATTN_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime(); unsigned long long trace_t1 = 0
#define ATTN_TRACE_LOOP_END() trace_t1 = __builtin_amdgcn_s_memrealtime()
#define ATTN_TRACE_END(TAG)                                                                                         \
  if (g_attn_trace && threadIdx.x == 0) {                                                                           \
    unsigned long long* r = g_attn_trace + ((size_t)((TAG) ? 8192 : 0) + blockIdx.x) * 4; /* [fwd 8192 | bwd 8192] */ \
    r[0] = trace_t0; r[1] = trace_t1; r[2] = __builtin_amdgcn_s_memrealtime();                                      \
    r[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | ((unsigned long long)(TAG) << 48); \
  }
#else
#define ATTN_TRACE_BEGIN()
#define ATTN_TRACE_LOOP_END()
#define ATTN_TRACE_END(TAG)
#endif

// Diagnostic build only (-DVBX_ATTN_STEPTRACE, tools/attn_fwd_steptrace.sh): eight s_memtime stamps (100 MHz) per key tile and wave of the
// forward kernel -- 0 step entry, 1 behind the barrier and the DMA issue, 2 first K fragments arrived, 3 / 4 first / second S chain issued, 5 softmax done,
// 6 / 7 first / second P.V block issued; the time between consecutive stamps is summed over the key loop in scalar registers
// and written once per wave: [workgroup][wave][8 segment sums (segment 0 = stamp 7 -> next stamp 0), first stamp, last stamp].
#ifdef VBX_ATTN_STEPTRACE
__device__ unsigned long long* g_attn_steptrace = nullptr;
#define ATTN_ST_DECL() unsigned long long st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev_ = __builtin_amdgcn_s_memtime(), st_first_ = st_prev_
#define ATTN_ST(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc_[i] += t_ - st_prev_; st_prev_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define ATTN_ST_FLUSH()                                                                                        \
  if (g_attn_steptrace && (threadIdx.x & 63) == 0) {                                                           \
    unsigned long long* r_ = g_attn_steptrace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10;            \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) r_[i_] = st_acc_[i_];                                     \
    r_[8] = st_first_; r_[9] = st_prev_;                                                                       \
  }
#else
#define ATTN_ST_DECL()
#define ATTN_ST(i)
#define ATTN_ST_FLUSH()
#endif

// (Rounds 1-2's forward kernels -- the register-staged double buffer and the 3-slot / 3-per-CU LDS-DMA form "v2" -- were removed in
//  round 6; their measurements are in docs/history.md.  The helpers of the LDS-DMA ring they introduced follow.)

// ============================================================================ forward, v2: LDS-DMA ring
// K|V tiles (16 KiB per 64 keys) arrive by global_load_lds into a 3-slot ring (48 KiB -> 3 workgroups per CU) with two
// tiles in flight while one is consumed; no staging registers, one raw barrier per tile.  The DMA image is lane-linear,
// so the 16-byte XOR swizzle is applied to the per-lane SOURCE address (slot s holds logical chunk (s&7)^(row&7)).
// Fragment reads are inline asm (a compiler-visible ds_read would make hipcc drain vmcnt(0) while DMAs are in flight)
// with the ring slot / key block folded into the DS immediate.
constexpr int ASTAGE = 2 * TILE16;

VBX_DEV unsigned lds_addr32(const char* p) { return (unsigned)(size_t)LDS_PTR(char, p); }

VBX_DEV void dma_tile(char* dst, const u16* __restrict__ base, int row0, int row_lim, int tid) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int s = i * 256 + tid;
    const int row = s >> 3, c = (s & 7) ^ attn_swz(row);
    const int gr = min(row0 + row, row_lim - 1);  // rows past the end re-read the last row (masked / never stored)
    const u16* src = base + (long)gr * 64 + c * 8;
    char* wave_dst = dst + (i * 256 + (tid & ~63)) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
  }
}

// One LDS-DMA piece (1 KiB per wave-instruction) with a SCALAR 64-bit base and a 32-bit per-lane byte offset: no vector ALU work per
// piece.  m0 (the LDS destination of the wave) is compiler-reserved and not preserved around an asm statement, so it is saved,
// written and restored inside the statement that uses it (cdna_hip_programming.md 5.7).  Base and destination must come from
// scalar ALU code (an SGPR written by v_readfirstlane right before the statement would need five wait states).
VBX_DEV void glds16_sbase(unsigned voff, const u16* __restrict__ sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// max over the two half-waves (lane ^ 32) on the vector ALU.  Both results of the swap pass through an opaque asm before they are
// combined: hipcc (ROCm 7.2) folds op(r[0], r[1]) of a permlane32_swap(x, x) to r[0] otherwise.
VBX_DEV float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  unsigned r0 = r[0], r1 = r[1];
  asm volatile("" : "+v"(r0), "+v"(r1));
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}

template <int OFF>
VBX_DEV f16x8 asm_read_b128(unsigned a) {
  f16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF) : "memory");
  return r;
}
template <int OFF>
VBX_DEV void asm_read_tr(s16x4& lo, s16x4& hi, unsigned a, unsigned a8) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "i"(OFF) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a8), "i"(OFF) : "memory");
}

// ============================================================================ forward, v3: 4 workgroups per CU
// Same data flow as v2, resized so that FOUR workgroups fit a CU (<= 128 VGPRs, 2-slot ring = 32 KiB): the benchmark
// grid is 8 x 16 heads x 9 query tiles = 1152 workgroups, i.e. 1.5 rounds of the 768 slots v2 gets (measured: two
// ~35 us rounds, shader clock 1.8 GHz); with 1024 slots all full tiles run in ONE round.  One K|V tile is in flight
// while one is consumed (K/V panels are L2 hits thanks to the XCD-aware order, the deeper ring bought nothing).
// Register diet: K fragments in two halves, V^T fragments per 32-key block.
constexpr int A3ST = 2;
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel_v3(const u16* __restrict__ q16, const u16* __restrict__ k16,
                                                             const u16* __restrict__ vv, const uint8_t* __restrict__ mask,
                                                             u16* __restrict__ out, u16* __restrict__ outb,
                                                             float* __restrict__ lse, int H, int Np, float scale2, int BH, int xmap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // ring: [slot][K tile | V tile]
#include "attn_fwd_v3_body.inc"
}
// the same body with attention dropout (training only; three workgroups per CU: the keep words and their selects need registers)
#define VBX_FWD_DROP
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel_v3_drop(const u16* __restrict__ q16, const u16* __restrict__ k16,
                                                                  const u16* __restrict__ vv, const uint8_t* __restrict__ mask,
                                                                  u16* __restrict__ out, u16* __restrict__ outb,
                                                                  float* __restrict__ lse, int H, int Np, float scale2, int BH, int xmap,
                                                                  const unsigned* __restrict__ dbits, int W2, float rkeep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_DROP

#ifdef VBX_ATTN_DIAG  // tools/attn_ablation.sh: timing ablations of the forward step, separate instantiations of the same body
#define VBX_ABL_KERNEL(N)                                                                                                        \
  __global__ __launch_bounds__(256, 4) void attn_fwd_kernel_v3_abl##N(const u16* __restrict__ q16, const u16* __restrict__ k16,  \
                                                                      const u16* __restrict__ vv, const uint8_t* __restrict__ mask, \
                                                                      u16* __restrict__ out, u16* __restrict__ outb,           \
                                                                      float* __restrict__ lse, int H, int Np, float scale2, int BH, int xmap)
#define VBX_FWD_ABL 1
VBX_ABL_KERNEL(1) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 6
VBX_ABL_KERNEL(6) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 8
VBX_ABL_KERNEL(8) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 14
VBX_ABL_KERNEL(14) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 16
VBX_ABL_KERNEL(16) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 49
VBX_ABL_KERNEL(49) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 64
VBX_ABL_KERNEL(64) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 384
VBX_ABL_KERNEL(384) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 63
VBX_ABL_KERNEL(63) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#define VBX_FWD_ABL 447
VBX_ABL_KERNEL(447) { extern __shared__ __attribute__((aligned(16))) char smem[];
#include "attn_fwd_v3_body.inc"
}
#undef VBX_FWD_ABL
#endif

// ---- Round-4 experiments on the forward, measured at the benchmark grid (8 x 16 heads x 1040 rows, stand-alone, same box) and removed:
//  * "v4": 256-row workgroups, a wave owning TWO 32-query blocks whose chains are interleaved in one instruction stream (S of block B
//    beside the softmax of block A, P.V of A beside the softmax of B; K / V^T fragments read once for both; 3-slot ring with counted
//    vmcnt; two workgroups per CU = the 512 full tiles in ONE round).  238-244 VGPRs, no spills, correct (all attention tests).
//    72-76 us against v3's 67-69 us: with two waves per SIMD a 32-MFMA wave-step takes ~3800 cycles -- neither the in-wave interleave
//    (sched_group_barrier patterns 1 MFMA : 12 VALU + 4 TRANS, 1 : 20, or the compiler's own order) nor the halved LDS traffic buys
//    back what four independent waves per SIMD hide.
//  * plain instead of packed f32 VALU in the softmax (v_pk_fma / v_pk_add / v_pk_mul -> scalar, by source and by -fno-slp-vectorize
//    for the whole file): v3 68.8 vs 69.1 us (no change), the two-body backward 180.7 vs 178.4 us (12 spilled registers), v4 75.9 vs
//    72.1 us -- packed math is NOT the anti-lever here that it is in a one-wave-per-SIMD stream; instruction count is what counts.
//  * a compiler trap found on the way (hipcc / ROCm 7.2): op(r[0], r[1]) of r = __builtin_amdgcn_permlane32_swap(x, x, ..) is folded to
//    r[0] when nothing else uses the pair (the partner half-wave's value silently vanishes); pass both results through an opaque
//    asm("" : "+v"(r0), "+v"(r1)) before combining them.

// (Round 1 wrote a "ragged tile" role for this kernel -- the 128 workgroups per launch whose query tile holds only the 16
//  register-token rows split their KEYS over the four waves with private tiles straight from global memory.  Round 2 ran it:
//  correct (tests/test_ops_gpu.py -k attn_fwd), but the 128-forward sample got SLOWER, 349.8 -> 357.1 ms in the same run, so it
//  was removed.)

// ============================================================================ backward: delta
// delta[b,h,n] = sum_d dO[b,n,h*64+d] * O[b,n,h*64+d]
// (Round 2 tried folding this pass into the dq kernel's prologue -- the lane already holds half of its query's dO row, the
//  matching half of O is four more 16-byte loads and one lane^32 exchange.  Correct, and the train step got 0.15 ms SLOWER in the
//  same run (10.61 -> 10.77 ms): the extra per-lane loads lengthen every dq workgroup's serial prologue by more than the 9 us
//  launch they replace.  Kept as a separate streaming pass at 4 TB/s.)
template <bool O_F16>
__global__ void attn_delta_kernel(const u16* __restrict__ o, const u16* __restrict__ dout, float* __restrict__ delta, int H,
                                  int Np, long total_chunks) {
  const long c = blockIdx.x * (long)blockDim.x + threadIdx.x;  // one 8-element chunk per thread
  const long cc = min(c, total_chunks - 1);
  const uint4 a = *reinterpret_cast<const uint4*>(o + cc * 8);
  const uint4 g = *reinterpret_cast<const uint4*>(dout + cc * 8);
  const unsigned aw[4] = {a.x, a.y, a.z, a.w}, gw[4] = {g.x, g.y, g.z, g.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const float o0 = O_F16 ? f16_to_f32((u16)(aw[i] & 0xffff)) : bf16_to_f32((u16)(aw[i] & 0xffff));
    const float o1 = O_F16 ? f16_to_f32((u16)(aw[i] >> 16)) : bf16_to_f32((u16)(aw[i] >> 16));
    s += o0 * bf16_to_f32((u16)(gw[i] & 0xffff));
    s += o1 * bf16_to_f32((u16)(gw[i] >> 16));
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (c < total_chunks && (c & 7) == 0) {
    const long rowhead = c >> 3;  // (b*Np + n)*H + h
    const int hh = (int)(rowhead % H);
    const long bn = rowhead / H;
    const int n = (int)(bn % Np);
    const long bb = bn / Np;
    delta[(bb * H + hh) * Np + n] = s;
  }
}

// (Round 1's register-staged backward kernels, attn_bwd_dq_kernel / attn_bwd_dkdv_kernel, were removed in round 6: docs/history.md.)

// ---- Round-2 experiment, removed after measurement: "v2" backward kernels with 8 waves = two groups of four that work on
// alternate 32-row blocks of the streamed operand, one barrier interval apart (phase A: softmax VALU + the LDS fragment reads of
// the next phase; phase B: 16 / 12 MFMAs), tiles by LDS-DMA into a 4-slot ring, zero-page rows instead of masks -- the structure
// that took the GEMM k-loop (gemm3.hip) from 700 to 1290 TFLOP/s.  Both kernels were correct on the unmasked tests and SLOWER in
// the train step (same run): dk/dv 117 -> ~163 us, dq 93 -> ~124 us.  Why: (1) 100 KiB of ring means ONE workgroup per CU, so
// the per-workgroup prologue (per-lane K / V fragments from global, first tiles) and epilogue (merge of the two groups through
// LDS, fused qk-norm / rotary backward, stores) of 4.5 workgroups per CU run back to back with nothing to overlap them, where v1's
// two independent workgroups per CU hide each other's; (2) the loop itself cannot be shortened much: every 64-row tile step
// stages 24 KiB per workgroup, 1152 x 17 x 24 KiB = 480 MB of L2 -> LDS traffic per launch, i.e. ~48 us at the ~10 TB/s fill
// rate all of this library's LDS-DMA loops top out at (gemm3.hip) -- v1 is within 2-2.4x of that floor, not of the MFMA peak.
// What would move it: 256 owned rows per workgroup (halves the staged bytes per MFMA; Np = 1040 then wastes 19 % of the slots).
// ============================================================================ backward: dk, dv with an LDS-DMA ring
// v1's decomposition and serial per-wave order (4 waves x 32 keys, 64-row query tiles, both 32-row blocks of a tile per wave), but
// the Q16 | Qb | dO tiles and the L / delta rows arrive by LDS-DMA into a 2-slot ring instead of through 48 staging registers:
// <= 168 VGPRs and 49 KiB of LDS -> THREE workgroups per CU instead of two (1152 workgroups = 1.5 rounds instead of 2.25).
// An invalid / masked key only affects its own dK / dV row, which is zeroed once after the loop: the loop carries no masks.  All LDS reads
// are inline asm (a compiler-visible ds_read would make hipcc drain the DMA queue with vmcnt(0) at every read).
constexpr int D3_SLOT = 3 * TILE16 + 512;  // Q16 | Qb | dO | L[64] | delta[64]
constexpr int D3_LDS = 2 * D3_SLOT;        // 50176 B
__device__ const float attn_big_page[4] = {1e30f, 1e30f, 1e30f, 1e30f};  // L of a row past Np

#define D3_READ128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define D3_READTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")

// DROP (training-time attention dropout): dbits = column-major keep bits C[bh][key][W2] (ops.hip).  dV sees the dropped
// probabilities (its 1 / keep is applied once to the accumulator), dS = P * (keep_bit * dP / keep - delta) the undropped ones.
template <bool DROP>
__device__ __forceinline__ void attn_bwd_dkdv_dma_body(char* smem, int wg_id, const u16* __restrict__ q16,
                                                       const u16* __restrict__ qb16, const u16* __restrict__ k16,
                                                       const u16* __restrict__ vv, const uint8_t* __restrict__ mask,
                                                       const u16* __restrict__ dout, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, float* __restrict__ dk,
                                                       u16* __restrict__ dv, int dv_ld, int H, int Np, float scale2, float scale,
                                                       int BH, int xmap, const QKBwd& fk, const unsigned* __restrict__ dbits,
                                                       int W2, float rkeep) {
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const AttnCoord co = attn_coord_id(wg_id, H, Np, BH, xmap);
  if (!co.ok) return;
  ATTN_TRACE_BEGIN();
  const int h = co.h, b = co.b;
  const long bh = (long)b * H + h;
  const u16* qbase = q16 + bh * Np * 64;
  const u16* qbbase = qb16 + bh * Np * 64;
  const u16* dobase = dout + (long)b * Np * (H * 64) + h * 64;
  const int key0 = co.tile * 128 + wave * 32;
  const bool active = key0 < Np;
  const int key = key0 + (lane & 31);
  const int keyc = min(key, Np - 1);
  bool kvalid = key < Np;
  if (kvalid && mask) kvalid = mask[(long)b * Np + key] != 0;
  const int ntiles = (Np + 63) / 64;

  f16x8 kf[4];
  bf16x8 vf[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    kf[t] = *reinterpret_cast<const f16x8*>(k16 + (bh * Np + keyc) * 64 + 16 * t + 8 * hi);
    vf[t] = *reinterpret_cast<const bf16x8*>(vv + (bh * Np + keyc) * 64 + 16 * t + 8 * hi);
  }
#pragma unroll
  for (int t = 0; t < 4; t++) {  // retire the per-lane fragment loads before any LDS-DMA is in flight
    asm volatile("" ::"v"(kf[t]));
    asm volatile("" ::"v"(vf[t]));
  }
  asm volatile("" ::"v"(kvalid));

  // Rows past Np are clamped to the last row (finite data) and get L = 1e30 from a constant page, so P = exp2(S - 1e30) = 0 and
  // every contribution of such a row vanishes.  Addresses are uniform base + 32-bit per-thread offset: nothing 64-bit stays live.
  auto issue = [&](int qt) {  // 2 pieces per thread per tile + the statistics rows (waves 0 / 1)
    char* slot = smem + (qt & 1) * D3_SLOT;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int sidx = i * 256 + tid;
      const int row = sidx >> 3, c = (sidx & 7) ^ attn_swz(row);
      const int gr = min(qt * 64 + row, Np - 1);
      const unsigned oq = (unsigned)(gr * 64 + c * 8) * 2u, od = (unsigned)(gr * (H * 64) + c * 8) * 2u;
      const char* s0 = reinterpret_cast<const char*>(qbase) + oq;
      const char* s1 = reinterpret_cast<const char*>(qbbase) + oq;
      const char* s2 = reinterpret_cast<const char*>(dobase) + od;
      char* wdst = slot + (i * 256 + wave * 64) * 16;  // wave-uniform; the DMA adds lane * 16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0, (__attribute__((address_space(3))) void*)wdst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1, (__attribute__((address_space(3))) void*)(wdst + TILE16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s2, (__attribute__((address_space(3))) void*)(wdst + 2 * TILE16), 16, 0, 0);
    }
    if (wave < 2) {  // wave 0: L, wave 1: delta -- 64 floats, lane-linear
      const int n = qt * 64 + lane;
      const float* src = (wave == 0 ? lse : delta) + bh * Np + min(n, Np - 1);
      if (wave == 0 && n >= Np) src = attn_big_page;
      char* sdst = slot + 3 * TILE16 + wave * 256;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)sdst, 4, 0, 0);
    }
  };

  // per-lane LDS addresses inside slot 0 / block 0 (slot, tile, block, t2, row-group offsets are DS immediates)
  unsigned ra[4], ta[2], ta8[2], sa;
  {
    const int row = lane & 31;
    const int G = lane >> 4, a16 = lane & 15;
    const int trow = 4 * (G >> 1) + (a16 >> 2);
#pragma unroll
    for (int t = 0; t < 4; t++) ra[t] = lds_addr32(smem + swz_off2(row, 2 * t + hi));
#pragma unroll
    for (int db = 0; db < 2; db++) {
      const int d = db * 32 + (G & 1) * 16 + 4 * (a16 & 3);
      ta[db] = lds_addr32(smem + swz_off2(trow, d >> 3) + (d & 7) * 2);
      ta8[db] = lds_addr32(smem + swz_off2(trow + 8, d >> 3) + (d & 7) * 2);
    }
    sa = lds_addr32(smem + 3 * TILE16 + 4 * hi * 4);
  }

  f32x16 adk[2], adv[2];
#pragma unroll
  for (int i = 0; i < 16; i++) { adk[0][i] = 0.f; adk[1][i] = 0.f; adv[0][i] = 0.f; adv[1][i] = 0.f; }

  const unsigned* bcol = DROP ? dbits + (bh * Np + keyc) * W2 : nullptr;
  uint2 wkeep = make_uint2(0u, 0u);  // keep bits of this lane's key against the 64 queries of the current tile
  if (DROP) wkeep = *reinterpret_cast<const uint2*>(bcol);
  issue(0);
  // one 32-row block QB of the tile in slot SO (both compile time)
  auto block = [&](auto so_c, auto qb_c) {
    constexpr int SO = decltype(so_c)::value, QB = decltype(qb_c)::value;
    constexpr int O = SO + QB * 4096;
    f32x16 s, dp;
    {
      f16x8 qfr[4];
      D3_READ128(qfr[0], ra[0], O); D3_READ128(qfr[1], ra[1], O); D3_READ128(qfr[2], ra[2], O); D3_READ128(qfr[3], ra[3], O);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = 0.f;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 4; t++) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(qfr[t], kf[t], s, 0, 0, 0);  // S[q][key] = Q . K^T
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      bf16x8 dofr[4];
      D3_READ128(dofr[0], ra[0], O + 2 * TILE16); D3_READ128(dofr[1], ra[1], O + 2 * TILE16);
      D3_READ128(dofr[2], ra[2], O + 2 * TILE16); D3_READ128(dofr[3], ra[3], O + 2 * TILE16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; e++) dp[e] = 0.f;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 4; t++) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dofr[t], vf[t], dp, 0, 0, 0);  // dP = dO . V^T
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      f32x4 l4[4];
#if defined(VBX_ATTN_ABL_DKDV) && (VBX_ATTN_ABL_DKDV & 1)  // timing ablation (diagnostic builds only): no statistics reads
      for (int g4 = 0; g4 < 4; g4++) l4[g4] = f32x4{scale, scale, scale, scale};
#else
      D3_READ128(l4[0], sa, SO + QB * 128); D3_READ128(l4[1], sa, SO + QB * 128 + 32);
      D3_READ128(l4[2], sa, SO + QB * 128 + 64); D3_READ128(l4[3], sa, SO + QB * 128 + 96);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#if defined(VBX_ATTN_ABL_DKDV) && (VBX_ATTN_ABL_DKDV & 2)  // timing ablation: no exponentials
          s[4 * g4 + j] = fmaf(s[4 * g4 + j], scale2, -l4[g4][j]);
#else
          s[4 * g4 + j] = fast_exp2(fmaf(s[4 * g4 + j], scale2, -l4[g4][j]));
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      f32x4 d4[4];
#if defined(VBX_ATTN_ABL_DKDV) && (VBX_ATTN_ABL_DKDV & 1)
      for (int g4 = 0; g4 < 4; g4++) d4[g4] = f32x4{scale2, scale2, scale2, scale2};
#else
      D3_READ128(d4[0], sa, SO + QB * 128 + 256); D3_READ128(d4[1], sa, SO + QB * 128 + 288);
      D3_READ128(d4[2], sa, SO + QB * 128 + 320); D3_READ128(d4[3], sa, SO + QB * 128 + 352);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DROP) {
        const unsigned wk = (QB ? wkeep.y : wkeep.x) >> (4 * hi);  // register 4 * g4 + j <-> query 8 * g4 + 4 * hi + j of the block
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const bool keep = (wk >> (8 * g4 + j)) & 1u;
            dp[4 * g4 + j] = s[4 * g4 + j] * ((keep ? dp[4 * g4 + j] * rkeep : 0.f) - d4[g4][j]);
            if (!keep) s[4 * g4 + j] = 0.f;
          }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++)
#pragma unroll
          for (int j = 0; j < 4; j++) dp[4 * g4 + j] = s[4 * g4 + j] * (dp[4 * g4 + j] - d4[g4][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 pf[2], dsf[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; t2++) {
      pf[t2] = pack_frag(s, t2);
      dsf[t2] = pack_frag(dp, t2);
    }
    __builtin_amdgcn_sched_barrier(0);
    s16x4 tl[4], th[4];  // transposed dO / Qb fragments of rows +0..15 (t2 = 0)
    D3_READTR(tl[0], ta[0], O + 2 * TILE16); D3_READTR(th[0], ta8[0], O + 2 * TILE16);
    D3_READTR(tl[1], ta[1], O + 2 * TILE16); D3_READTR(th[1], ta8[1], O + 2 * TILE16);
    D3_READTR(tl[2], ta[0], O + TILE16); D3_READTR(th[2], ta8[0], O + TILE16);
    D3_READTR(tl[3], ta[1], O + TILE16); D3_READTR(th[3], ta8[1], O + TILE16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const s16x8 v8 = {tl[j][0], tl[j][1], tl[j][2], tl[j][3], th[j][0], th[j][1], th[j][2], th[j][3]};
      fr[j] = __builtin_bit_cast(bf16x8, v8);
    }
    __builtin_amdgcn_s_setprio(1);
    adv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], pf[0], adv[0], 0, 0, 0);
    adk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2], dsf[0], adk[0], 0, 0, 0);
    adv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], pf[0], adv[1], 0, 0, 0);
    adk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[3], dsf[0], adk[1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // rows +16..31 (t2 = 1): 2048 bytes further, same swizzle phase
    D3_READTR(tl[0], ta[0], O + 2 * TILE16 + 2048); D3_READTR(th[0], ta8[0], O + 2 * TILE16 + 2048);
    D3_READTR(tl[1], ta[1], O + 2 * TILE16 + 2048); D3_READTR(th[1], ta8[1], O + 2 * TILE16 + 2048);
    D3_READTR(tl[2], ta[0], O + TILE16 + 2048); D3_READTR(th[2], ta8[0], O + TILE16 + 2048);
    D3_READTR(tl[3], ta[1], O + TILE16 + 2048); D3_READTR(th[3], ta8[1], O + TILE16 + 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const s16x8 v8 = {tl[j][0], tl[j][1], tl[j][2], tl[j][3], th[j][0], th[j][1], th[j][2], th[j][3]};
      fr[j] = __builtin_bit_cast(bf16x8, v8);
    }
    __builtin_amdgcn_s_setprio(1);
    adv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], pf[1], adv[0], 0, 0, 0);
    adk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2], dsf[1], adk[0], 0, 0, 0);
    adv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], pf[1], adv[1], 0, 0, 0);
    adk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[3], dsf[1], adk[1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](auto so_c, int qt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's pieces of tile qt have landed
    __builtin_amdgcn_s_barrier();                     // tile qt visible to all; everyone is done with tile qt-1
    __builtin_amdgcn_sched_barrier(0);
    if (qt + 1 < ntiles) issue(qt + 1);
    if (!active) return;
    block(so_c, std::integral_constant<int, 0>{});
    if (qt * 64 + 32 < Np) block(so_c, std::integral_constant<int, 1>{});
    if (DROP && qt + 1 < ntiles) wkeep = *reinterpret_cast<const uint2*>(bcol + 2 * (qt + 1));  // retired by the next step's vmcnt(0)
  };
  for (int qt = 0; qt < ntiles; qt += 2) {
    step(std::integral_constant<int, 0>{}, qt);
    if (qt + 1 < ntiles) step(std::integral_constant<int, D3_SLOT>{}, qt + 1);
  }
  __syncthreads();  // every wave is done with the ring before it becomes epilogue staging space
  ATTN_TRACE_LOOP_END();

  if (active && !kvalid) {  // out-of-range or masked key: its softmax weight is 0 for every query
#pragma unroll
    for (int e = 0; e < 16; e++) { adk[0][e] = 0.f; adk[1][e] = 0.f; adv[0][e] = 0.f; adv[1][e] = 0.f; }
  }
  if constexpr (DROP) {
#pragma unroll
    for (int e = 0; e < 16; e++) { adv[0][e] *= rkeep; adv[1][e] *= rkeep; }
  }
  if (fk.dqkv)  // fused rotary + qk-norm backward of dk (all waves: it ends with workgroup barriers)
    store_rows_qknorm(smem + wave * 12288, reinterpret_cast<float*>(smem + 4 * 12288), adk, scale, fk, active, b, h, H, co.tile,
                      (Np + 127) >> 7, key0, Np, lane, wave);
  if (active) {
    char* wst = smem + wave * 12288;  // 8 KiB fp32 dk block | 4 KiB bf16 dv block
    if (!fk.dqkv) store_rows_f32(wst, adk, scale, dk + bh * Np * 64, key0, Np, lane);
    char* vst = wst + 8192;
    const int kl = lane & 31;
#pragma unroll
    for (int db = 0; db < 2; db++)
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int d = db * 32 + 8 * g4 + 4 * hi;
        *reinterpret_cast<uint2*>(vst + kl * 128 + (((d >> 3) ^ (kl & 7)) << 4) + (d & 7) * 2) =
            make_uint2(pack_bf16x2(adv[db][4 * g4 + 0], adv[db][4 * g4 + 1]), pack_bf16x2(adv[db][4 * g4 + 2], adv[db][4 * g4 + 3]));
      }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = it * 8 + (lane >> 3), ch = lane & 7;
      if (key0 + row < Np)
        *reinterpret_cast<uint4*>(dv + ((long)b * Np + key0 + row) * dv_ld + h * 64 + ch * 8) =
            *reinterpret_cast<const uint4*>(vst + row * 128 + ((ch ^ (row & 7)) << 4));
    }
  }
  ATTN_TRACE_END(2);
}

// ---------------------------------------------------------------------------- backward: dq with the same ring
// v1's decomposition (4 waves x 32 queries, 64-key tiles of K16 | Kb | V) with LDS-DMA staging and asm fragment reads: no staging
// registers -> three workgroups per CU.  Key masks (user mask / keys past Np) are applied per element only in tiles that need them,
// exactly as the forward does.
// DROP: dbits = row-major keep bits R[bh][q][W2], as in the forward.
constexpr int Q3_SLOT = 3 * TILE16;
template <bool DROP>
__device__ __forceinline__ void attn_bwd_dq_dma_body(char* smem, int wg_id, const u16* __restrict__ q16,
                                                     const u16* __restrict__ k16, const u16* __restrict__ kb16,
                                                     const u16* __restrict__ vv, const uint8_t* __restrict__ mask,
                                                     const u16* __restrict__ dout, const float* __restrict__ lse,
                                                     const float* __restrict__ delta, float* __restrict__ dq, int H, int Np,
                                                     float scale2, float scale, int BH, int xmap, const QKBwd& fq,
                                                     const unsigned* __restrict__ dbits, int W2, float rkeep) {
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const AttnCoord co = attn_coord_id(wg_id, H, Np, BH, xmap);
  if (!co.ok) return;
  ATTN_TRACE_BEGIN();
  const int h = co.h, b = co.b;
  const long bh = (long)b * H + h;
  const u16* kbase = k16 + bh * Np * 64;
  const u16* kbbase = kb16 + bh * Np * 64;
  const u16* vbase = vv + bh * Np * 64;
  const int q0 = co.tile * 128 + wave * 32;
  const bool active = q0 < Np;
  const int q = q0 + (lane & 31);
  const int qc = min(q, Np - 1);
  const int ntiles = (Np + 63) / 64;

  f16x8 qf[4];
  bf16x8 dof[4];
#pragma unroll
  for (int t = 0; t < 4; t++) {
    qf[t] = *reinterpret_cast<const f16x8*>(q16 + (bh * Np + qc) * 64 + 16 * t + 8 * hi);
    dof[t] = *reinterpret_cast<const bf16x8*>(dout + ((long)b * Np + qc) * (H * 64) + h * 64 + 16 * t + 8 * hi);
  }
  const float L2 = lse[bh * Np + qc];
  const float dlt = delta[bh * Np + qc];
#pragma unroll
  for (int t = 0; t < 4; t++) {  // retire the per-lane loads before any LDS-DMA is in flight
    asm volatile("" ::"v"(qf[t]));
    asm volatile("" ::"v"(dof[t]));
  }
  asm volatile("" ::"v"(L2), "v"(dlt));
  const unsigned* brow = DROP ? dbits + (bh * Np + qc) * W2 : nullptr;
  uint2 wkeep = make_uint2(0u, 0u);
  if (DROP) wkeep = *reinterpret_cast<const uint2*>(brow);

  auto issue = [&](int kt) {
    char* slot = smem + (kt & 1) * Q3_SLOT;
    dma_tile(slot, kbase, kt * 64, Np, tid);
    dma_tile(slot + TILE16, kbbase, kt * 64, Np, tid);
    dma_tile(slot + 2 * TILE16, vbase, kt * 64, Np, tid);
  };
  issue(0);

  unsigned ra[4], ta[2], ta8[2];
  {
    const int row = lane & 31;
    const int G = lane >> 4, a16 = lane & 15;
    const int trow = 4 * (G >> 1) + (a16 >> 2);
#pragma unroll
    for (int t = 0; t < 4; t++) ra[t] = lds_addr32(smem + swz_off2(row, 2 * t + hi));
#pragma unroll
    for (int db = 0; db < 2; db++) {
      const int d = db * 32 + (G & 1) * 16 + 4 * (a16 & 3);
      ta[db] = lds_addr32(smem + TILE16 + swz_off2(trow, d >> 3) + (d & 7) * 2);
      ta8[db] = lds_addr32(smem + TILE16 + swz_off2(trow + 8, d >> 3) + (d & 7) * 2);
    }
  }

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; i++) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

  auto block = [&](auto so_c, auto kb_c, int k0, bool need_mask) {
    constexpr int SO = decltype(so_c)::value, KB = decltype(kb_c)::value;
    constexpr int O = SO + KB * 4096;
    f32x16 s, dp;
    f16x8 kfr[4];
    bf16x8 vfr[4];
    D3_READ128(kfr[0], ra[0], O); D3_READ128(kfr[1], ra[1], O); D3_READ128(kfr[2], ra[2], O); D3_READ128(kfr[3], ra[3], O);
    D3_READ128(vfr[0], ra[0], O + 2 * TILE16); D3_READ128(vfr[1], ra[1], O + 2 * TILE16);
    D3_READ128(vfr[2], ra[2], O + 2 * TILE16); D3_READ128(vfr[3], ra[3], O + 2 * TILE16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) { s[e] = 0.f; dp[e] = 0.f; }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 4; t++) {  // S[key][q] = K . Q^T ; dP[key][q] = V . dO^T   (two independent chains)
      s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfr[t], qf[t], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[t], dof[t], dp, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    // Kb^T fragments of keys +0..15: requested now, they land while the VALU block runs
    s16x4 tl[4], th[4];
    D3_READTR(tl[0], ta[0], O); D3_READTR(th[0], ta8[0], O);
    D3_READTR(tl[1], ta[1], O); D3_READTR(th[1], ta8[1], O);
    D3_READTR(tl[2], ta[0], O + 2048); D3_READTR(th[2], ta8[0], O + 2048);
    D3_READTR(tl[3], ta[1], O + 2048); D3_READTR(th[3], ta8[1], O + 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float pv = fast_exp2(fmaf(s[r], scale2, -L2));
      if (need_mask) {
        const int kg = k0 + KB * 32 + acc_row(r, hi);
        bool ok = kg < Np;
        if (ok && mask) ok = mask[(long)b * Np + kg] != 0;
        if (!ok) pv = 0.f;
      }
      if constexpr (DROP) {
        const bool keep = (((KB ? wkeep.y : wkeep.x) >> (4 * hi)) >> (8 * (r >> 2) + (r & 3))) & 1u;
        s[r] = pv * ((keep ? dp[r] * rkeep : 0.f) - dlt);
      } else {
        s[r] = pv * (dp[r] - dlt);
      }
    }
    const bf16x8 ds0 = pack_frag(s, 0), ds1 = pack_frag(s, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 fr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const s16x8 v8 = {tl[j][0], tl[j][1], tl[j][2], tl[j][3], th[j][0], th[j][1], th[j][2], th[j][3]};
      fr[j] = __builtin_bit_cast(bf16x8, v8);
    }
    __builtin_amdgcn_s_setprio(1);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], ds0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], ds0, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2], ds1, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[3], ds1, acc[1], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step = [&](auto so_c, int kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's pieces of tile kt have landed
    __builtin_amdgcn_s_barrier();                     // tile kt visible to all; everyone is done with tile kt-1
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < ntiles) issue(kt + 1);
    if (!active) return;
    const int k0 = kt * 64;
    const bool need_mask = (mask != nullptr) || (k0 + 64 > Np);
    block(so_c, std::integral_constant<int, 0>{}, k0, need_mask);
    if (k0 + 32 < Np) block(so_c, std::integral_constant<int, 1>{}, k0, need_mask);
    if (DROP && kt + 1 < ntiles) wkeep = *reinterpret_cast<const uint2*>(brow + 2 * (kt + 1));  // retired by the next step's vmcnt(0)
  };
  for (int kt = 0; kt < ntiles; kt += 2) {
    step(std::integral_constant<int, 0>{}, kt);
    if (kt + 1 < ntiles) step(std::integral_constant<int, Q3_SLOT>{}, kt + 1);
  }
  __syncthreads();  // every wave is done with the ring before it becomes epilogue staging space
  ATTN_TRACE_LOOP_END();

  if (fq.dqkv) {  // fused rotary + qk-norm backward -> bf16 d(qkv); the red scratch sits behind the four 8 KiB wave blocks
    store_rows_qknorm(smem + wave * 8192, reinterpret_cast<float*>(smem + 4 * 8192), acc, scale, fq, active, b, h, H, co.tile,
                      (Np + 127) >> 7, q0, Np, lane, wave);
    ATTN_TRACE_END(1);
    return;
  }
  if (active) store_rows_f32(smem + wave * 8192, acc, scale, dq + bh * Np * 64, q0, Np, lane);
  ATTN_TRACE_END(1);
}

// dq and dk/dv depend on the same inputs and not on each other: ONE launch carries both grids, so the chip goes through one
// drain phase (the last, partly filled round of workgroups) instead of two.  role: 0 both, 1 dq only, 2 dk/dv only (A/B, tests).
struct AttnBwdArgs {
  const u16 *q16, *k16, *qb16, *kb16, *vv, *dout;
  const uint8_t* mask;
  const float *lse, *delta;
  float *dq, *dk;
  u16* dv;
  int dv_ld, H, Np, BH, xmap, grid_one, role;
  float scale2, scale;
  QKBwd fq, fk;
  const unsigned *bits_rm, *bits_cm;  // attention dropout keep bits (ops.hip::attn_dropout_bits_kernel), DROP instantiation only
  int W2;
  float rkeep;
};
constexpr int BWD_DMA_LDS = D3_LDS > 2 * Q3_SLOT ? D3_LDS : 2 * Q3_SLOT;
template <bool DROP>
__global__ __launch_bounds__(256, 3) void attn_bwd_kernel_dma(const AttnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int id = blockIdx.x, role;
  if (a.role) {
    role = a.role == 1;
  } else {  // launch order: tails of both roles, dk/dv full tiles (the longer ones), dq full tiles
    const int T = attn_tail_ids(a.Np, a.BH, a.xmap), F = a.grid_one - T;
    if (a.xmap & 2) {  // A/B: tails last
      if (id < 2 * F) { role = id >= F; id -= role * F; }
      else { id -= 2 * F; role = id >= T; id += F - role * T; }
    } else if (id < 2 * T) {
      role = id >= T;
      id -= role * T;
    } else {
      id -= 2 * T;
      role = id >= F;
      id += T - role * F;
    }
  }
  if (role)
    attn_bwd_dq_dma_body<DROP>(smem, id, a.q16, a.k16, a.kb16, a.vv, a.mask, a.dout, a.lse, a.delta, a.dq, a.H, a.Np, a.scale2,
                               a.scale, a.BH, a.xmap, a.fq, a.bits_rm, a.W2, a.rkeep);
  else
    attn_bwd_dkdv_dma_body<DROP>(smem, id, a.q16, a.qb16, a.k16, a.vv, a.mask, a.dout, a.lse, a.delta, a.dk, a.dv, a.dv_ld, a.H,
                                 a.Np, a.scale2, a.scale, a.BH, a.xmap, a.fk, a.bits_cm, a.W2, a.rkeep);
}

#include "attn_bwd_fold.inc"

}  // namespace

#ifdef VBX_ATTN_TRACE
extern "C" int vbx_debug_attn_trace(void* buf) {  // diagnostic build only: buf = [2][8192][4] u64 (forward | backward launches, by blockIdx), null to stop
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

static const float LOG2E = 1.4426950408889634f;
// Workgroup order of the attention launches (attn_coord_id): bit 0 = XCD-local heads (always), bit 1 = the tail tiles LAST.  A tail of
// <= 16 rows runs the short 16 x 16-shaped role (round 6): dispatched first it holds 128 of the 1024 slots while the last full tiles wait;
// dispatched last it fills the drain of the launch -- same call, tails first -> last: train step 8.86 -> 8.73 ms, 64-interval sample
// 283.6 -> 279.6 ms (profiles/r06_ab_attn_tails_last.txt).  Longer tails (one to four waves walking the whole key loop at the pace of a
// full tile) still go first, where they overlap with full workgroups.  VBX_ATTN_XMAP=<bits> overrides.
static int attn_xmap_for(int Np) {
  static const int forced = getenv("VBX_ATTN_XMAP") ? atoi(getenv("VBX_ATTN_XMAP")) : -1;
  if (forced >= 0) return forced;
  const int tail = Np & 127;
  return (tail != 0 && tail <= 16) ? 3 : 1;
}
// Round 5 contract (include/vbx.h): q16 arrives PRE-MULTIPLIED by scale * log2(e), so q . k is already the exponent of exp2 and the
// kernels that still carry a `scale2` factor get 1; `scale` itself is only the multiplier of dq / dk.
static const float QK_UNIT = 1.0f;

extern "C" float vbx_dropout_keep_scale(float p);
extern "C" int vbx_dropout_bits_words(int Np);

static int attn_fwd_impl(const void* q16, const void* k16, const void* v, const uint8_t* mask, void* out, void* out_bf16,
                         float* lse, int B, int H, int Np, float scale, const void* drop_bits_rm, float drop_p, void* stream) {
  VBX_REQUIRE(q16 && k16 && v && out && lse, "vbx_attn_fwd: null pointer");
  VBX_REQUIRE(B > 0 && H > 0 && Np > 0 && scale > 0.f, "vbx_attn_fwd: bad dims");
  const int xmap = attn_xmap_for(Np);  // VBX_ATTN_XMAP=0: A/B against the plain tile order
  const int BH = B * H;
  dim3 grid(cdiv(Np, 128) * ((xmap & 1) ? cdiv(BH, 8) * 8 : BH));
  if (drop_bits_rm) {  // training-time attention dropout (attend.py:131): the 4-per-CU body with the keep-bit selects
    VBX_REQUIRE(drop_p > 0.f && drop_p < 1.f, "vbx_attn_fwd_dropout: p must be in (0, 1)");
    hipLaunchKernelGGL(attn_fwd_kernel_v3_drop, grid, dim3(256), A3ST * ASTAGE, (hipStream_t)stream, (const u16*)q16,
                       (const u16*)k16, (const u16*)v, mask, (u16*)out, (u16*)out_bf16, lse, H, Np, QK_UNIT, BH, xmap,
                       (const unsigned*)drop_bits_rm, vbx_dropout_bits_words(Np), vbx_dropout_keep_scale(drop_p));
    VBX_LAUNCH_CHECK();
    return 0;
  }
#ifdef VBX_ATTN_DIAG
  if (getenv("VBX_FWD_ABL3") && atoi(getenv("VBX_FWD_ABL3")) != 0) {
    const int a3 = atoi(getenv("VBX_FWD_ABL3"));
#define VBX_ABL_LAUNCH(N)                                                                                                  \
  case N:                                                                                                                  \
    hipLaunchKernelGGL(attn_fwd_kernel_v3_abl##N, grid, dim3(256), A3ST * ASTAGE, (hipStream_t)stream, (const u16*)q16,     \
                       (const u16*)k16, (const u16*)v, mask, (u16*)out, (u16*)out_bf16, lse, H, Np, QK_UNIT, BH, xmap);     \
    break;
    switch (a3) {
      VBX_ABL_LAUNCH(1) VBX_ABL_LAUNCH(6) VBX_ABL_LAUNCH(8) VBX_ABL_LAUNCH(14) VBX_ABL_LAUNCH(16) VBX_ABL_LAUNCH(49) VBX_ABL_LAUNCH(64)
      VBX_ABL_LAUNCH(384) VBX_ABL_LAUNCH(63) VBX_ABL_LAUNCH(447)
      default: VBX_REQUIRE(false, "VBX_FWD_ABL3: no such ablation build");
    }
#undef VBX_ABL_LAUNCH
    VBX_LAUNCH_CHECK();
    return 0;
  }
#endif
  hipLaunchKernelGGL(attn_fwd_kernel_v3, grid, dim3(256), A3ST * ASTAGE, (hipStream_t)stream, (const u16*)q16, (const u16*)k16,
                     (const u16*)v, mask, (u16*)out, (u16*)out_bf16, lse, H, Np, QK_UNIT, BH, xmap);
  VBX_LAUNCH_CHECK();
  return 0;
}
#ifdef VBX_ATTN_STEPTRACE
extern "C" int vbx_debug_attn_steptrace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_steptrace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" float vbx_attn_q_prescale(float scale) { return scale * LOG2E; }
extern "C" int vbx_attn_fwd(const void* q16, const void* k16, const void* v, const uint8_t* mask, void* out, void* out_bf16,
                            float* lse, int B, int H, int Np, float scale, void* stream) {
  return attn_fwd_impl(q16, k16, v, mask, out, out_bf16, lse, B, H, Np, scale, nullptr, 0.f, stream);
}
extern "C" int vbx_attn_fwd_dropout(const void* q16, const void* k16, const void* v, const uint8_t* mask, void* out, void* out_bf16,
                                    float* lse, int B, int H, int Np, float scale, const void* bits_rm, float p, void* stream) {
  VBX_REQUIRE(bits_rm, "vbx_attn_fwd_dropout: null keep bits");
  return attn_fwd_impl(q16, k16, v, mask, out, out_bf16, lse, B, H, Np, scale, bits_rm, p, stream);
}

// Backward variant: 0 / 1 = the two-body kernel with the softmax statistics folded into the MFMA accumulator (round 5, default),
// 3 = the same two bodies without the fold (round 3's arithmetic; also what attention dropout runs on).  Variant 2 was round 3's ONE-PASS
// chain kernel (every S / dP block evaluated once, dq summed over key blocks by an ordered chain of workgroups through device
// memory): correct, deterministic and 30 - 60 % slower on MI355X (232 - 262 us against 180 - 198 us; docs/history.md has its step-level
// time line and the stale-flag finding) -- removed in round 6.  vbx_attn_bwd_select(2) now returns VBX_EUNSUPPORTED,
// vbx_attn_bwd_scratch_bytes() 0 (no kernel needs scratch; the `scratch` arguments stay in the ABI and are ignored).
static int g_attn_bwd_variant = 0;
extern "C" int vbx_attn_bwd_select(int variant) {
  if (variant == 2) {
    vbx_set_error("vbx_attn_bwd_select: the one-pass kernel (2) was removed in round 6; 0 / 1 folded two-body, 3 unfolded two-body");
    return VBX_EUNSUPPORTED;
  }
  VBX_REQUIRE(variant >= 0 && variant <= 3, "vbx_attn_bwd_select: 0 auto, 1 two-body (folded), 3 two-body without the MFMA fold");
  g_attn_bwd_variant = variant;
  return 0;
}
extern "C" int vbx_attn_bwd_variant(void) { return 1; }
extern "C" size_t vbx_attn_bwd_scratch_bytes(int, int, int) { return 0; }

struct AttnDrop {  // training-time attention dropout: keep bits in both orientations (vbx_attn_dropout_bits) and the drop probability
  const unsigned *rm = nullptr, *cm = nullptr;
  float p = 0.f;
};
static int attn_bwd_impl(const void* q16, const void* k16, const void* qb, const void* kb, const void* v, const uint8_t* mask,
                         const void* out, int out_is_f16, const void* dout, const float* lse, float* delta, float* dq, float* dk,
                         void* dv, int dv_ld, int B, int H, int Np, float scale, const QKBwd& fq, const QKBwd& fk, void* scratch,
                         const AttnDrop& drop, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  static bool attr = false;
  const bool dropout = drop.rm != nullptr;
  VBX_REQUIRE(!dropout || (drop.cm && drop.p > 0.f && drop.p < 1.f), "vbx_attn_bwd_dropout: needs both keep-bit arrays and p in (0, 1)");
  (void)scratch;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel_dma<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_DMA_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel_dma<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_DMA_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel_fold), hipFuncAttributeMaxDynamicSharedMemorySize, BWD_DMA_LDS);
    attr = true;
  }
  const long chunks = (long)B * Np * H * 8;
  if (!out) {
    // delta was written by the to_out dgrad's epilogue (vbx_gemm_desc.delta): no pass of its own
  } else if (out_is_f16)
    hipLaunchKernelGGL(attn_delta_kernel<true>, dim3(cdiv(chunks, 256)), dim3(256), 0, st, (const u16*)out, (const u16*)dout,
                       delta, H, Np, chunks);
  else
    hipLaunchKernelGGL(attn_delta_kernel<false>, dim3(cdiv(chunks, 256)), dim3(256), 0, st, (const u16*)out, (const u16*)dout,
                       delta, H, Np, chunks);
  VBX_LAUNCH_CHECK();
  const int xmap = attn_xmap_for(Np);
  const int BH = B * H;
  dim3 grid(cdiv(Np, 128) * ((xmap & 1) ? cdiv(BH, 8) * 8 : BH));
  AttnBwdArgs a;
  a.bits_rm = drop.rm; a.bits_cm = drop.cm; a.W2 = vbx_dropout_bits_words(Np); a.rkeep = dropout ? vbx_dropout_keep_scale(drop.p) : 1.f;
  a.q16 = (const u16*)q16; a.k16 = (const u16*)k16; a.qb16 = (const u16*)qb; a.kb16 = (const u16*)kb; a.vv = (const u16*)v;
  a.dout = (const u16*)dout; a.mask = mask; a.lse = lse; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = (u16*)dv; a.dv_ld = dv_ld;
  a.H = H; a.Np = Np; a.BH = BH; a.xmap = xmap; a.grid_one = (int)grid.x; a.scale2 = QK_UNIT; a.scale = scale; a.fq = fq; a.fk = fk;
  a.role = 0;  // both bodies in ONE launch
  // VBX_ATTN_BWD_FOLD=0: A/B against round 3's bodies (statistics subtracted by VALU instead of folded into the MFMA accumulator)
  static const bool fold_env = !(getenv("VBX_ATTN_BWD_FOLD") && atoi(getenv("VBX_ATTN_BWD_FOLD")) == 0);
  const bool fold = fold_env && g_attn_bwd_variant != 3;
  if (dropout) hipLaunchKernelGGL(attn_bwd_kernel_dma<true>, dim3(2 * grid.x), dim3(256), BWD_DMA_LDS, st, a);
  else if (fold) hipLaunchKernelGGL(attn_bwd_kernel_fold, dim3(2 * grid.x), dim3(256), BWD_DMA_LDS, st, a);
  else hipLaunchKernelGGL(attn_bwd_kernel_dma<false>, dim3(2 * grid.x), dim3(256), BWD_DMA_LDS, st, a);
  VBX_LAUNCH_CHECK();
  return 0;
}

extern "C" int vbx_attn_bwd(const void* q16, const void* k16, const void* qb, const void* kb, const void* v,
                            const uint8_t* mask, const void* out, int out_is_f16, const void* dout, const float* lse,
                            float* delta, float* dq, float* dk, void* dv, int dv_ld, int B, int H, int Np, float scale,
                            void* scratch, void* stream) {
  VBX_REQUIRE(q16 && k16 && qb && kb && v && dout && lse && delta && dq && dk && dv, "vbx_attn_bwd: null pointer");
  VBX_REQUIRE(B > 0 && H > 0 && Np > 0 && scale > 0.f && dv_ld % 8 == 0, "vbx_attn_bwd: bad dims (dv_ld must be a multiple of 8)");
  const QKBwd none{};
  return attn_bwd_impl(q16, k16, qb, kb, v, mask, out, out_is_f16, dout, lse, delta, dq, dk, dv, dv_ld, B, H, Np, scale, none, none,
                       scratch, AttnDrop{}, stream);
}
extern "C" int vbx_attn_bwd_dropout(const void* q16, const void* k16, const void* qb, const void* kb, const void* v,
                                    const uint8_t* mask, const void* out, int out_is_f16, const void* dout, const float* lse,
                                    float* delta, float* dq, float* dk, void* dv, int dv_ld, int B, int H, int Np, float scale,
                                    const void* bits_rm, const void* bits_cm, float p, void* stream) {
  VBX_REQUIRE(q16 && k16 && qb && kb && v && dout && lse && delta && dq && dk && dv && bits_rm && bits_cm, "vbx_attn_bwd_dropout: null pointer");
  VBX_REQUIRE(B > 0 && H > 0 && Np > 0 && scale > 0.f && dv_ld % 8 == 0, "vbx_attn_bwd_dropout: bad dims (dv_ld must be a multiple of 8)");
  const QKBwd none{};
  return attn_bwd_impl(q16, k16, qb, kb, v, mask, out, out_is_f16, dout, lse, delta, dq, dk, dv, dv_ld, B, H, Np, scale, none, none,
                       nullptr, AttnDrop{(const unsigned*)bits_rm, (const unsigned*)bits_cm, p}, stream);
}

extern "C" int vbx_attn_bwd_fused_tiles(int Np) { return cdiv(Np, 128); }

extern "C" int vbx_attn_bwd_fused(const void* q16, const void* k16, const void* qb, const void* kb, const void* v,
                                  const uint8_t* mask, const void* out, int out_is_f16, const void* dout, const float* lse,
                                  float* delta, const float* q_rnorm, const float* k_rnorm, const float* q_gamma,
                                  const float* k_gamma, const float* rot_cos, const float* rot_sin, float qk_scale, void* dqkv,
                                  int ld, float* gpart, int B, int H, int Np, float scale, void* scratch, void* stream) {
  return vbx_attn_bwd_fused_dropout(q16, k16, qb, kb, v, mask, out, out_is_f16, dout, lse, delta, q_rnorm, k_rnorm, q_gamma, k_gamma,
                                    rot_cos, rot_sin, qk_scale, dqkv, ld, gpart, B, H, Np, scale, scratch, nullptr, nullptr, 0.f, stream);
}
extern "C" int vbx_attn_bwd_fused_dropout(const void* q16, const void* k16, const void* qb, const void* kb, const void* v,
                                          const uint8_t* mask, const void* out, int out_is_f16, const void* dout, const float* lse,
                                          float* delta, const float* q_rnorm, const float* k_rnorm, const float* q_gamma,
                                          const float* k_gamma, const float* rot_cos, const float* rot_sin, float qk_scale,
                                          void* dqkv, int ld, float* gpart, int B, int H, int Np, float scale, void* scratch,
                                          const void* bits_rm, const void* bits_cm, float p, void* stream) {
  VBX_REQUIRE(q16 && k16 && qb && kb && v && dout && lse && delta && dqkv && rot_cos && rot_sin, "vbx_attn_bwd_fused: null pointer");
  VBX_REQUIRE(B > 0 && H > 0 && Np > 0 && scale > 0.f && ld % 8 == 0 && ld >= 3 * H * 64, "vbx_attn_bwd_fused: bad dims");
  VBX_REQUIRE(qk_scale <= 0.f || (q_rnorm && k_rnorm && q_gamma && k_gamma && gpart), "vbx_attn_bwd_fused: qk-norm needs norms, gammas, gpart");
  const int I = H * 64, tiles = cdiv(Np, 128);
  QKBwd fq{(const u16*)q16, q_rnorm, q_gamma, rot_cos, rot_sin, qk_scale, (u16*)dqkv, ld, 0, gpart, 1.0f / (scale * LOG2E)};
  QKBwd fk{(const u16*)k16, k_rnorm, k_gamma, rot_cos, rot_sin, qk_scale, (u16*)dqkv, ld, I,
           gpart ? gpart + (size_t)B * tiles * H * 64 : nullptr, 1.0f};
  return attn_bwd_impl(q16, k16, qb, kb, v, mask, out, out_is_f16, dout, lse, delta, nullptr, nullptr, (u16*)dqkv + 2 * I, ld, B, H, Np,
                       scale, fq, fk, scratch, AttnDrop{(const unsigned*)bits_rm, (const unsigned*)bits_cm, p}, stream);
}
