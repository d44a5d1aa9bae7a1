// Register epilogues shared by the transposed-accumulator GEMM tiles (gemm3.hip: 256 x 256, gemm4.hip: 128 x 256).
#pragma once
#include "common.hpp"
#include "epi_stage_layout.hpp"

namespace gepi {

// Diagnostic build only (-DVBX_GEMM_TRACE, tools/build_trace_lib.sh): wave 0 of every workgroup records s_memrealtime (100 MHz) at
// kernel entry, after the prologue wait, after the k-loop and after the epilogue, plus where it ran -- tools/native/gemm_trace.cpp.
#ifdef VBX_GEMM_TRACE
static __device__ unsigned long long* g_gemm_trace = nullptr;  // one per translation unit (no RDC)
#define GEMM_TRACE_DECL() unsigned long long gtr0 = __builtin_amdgcn_s_memrealtime(), gtr1 = 0, gtr2 = 0
#define GEMM_TRACE_MARK(V) V = __builtin_amdgcn_s_memrealtime()
#define GEMM_TRACE_END()                                                                                          \
  if (gepi::g_gemm_trace && threadIdx.x == 0) {                                                                   \
    unsigned long long* r = gepi::g_gemm_trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 5;              \
    r[0] = gtr0; r[1] = gtr1; r[2] = gtr2; r[3] = __builtin_amdgcn_s_memrealtime();                              \
    r[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
  }
#else
#define GEMM_TRACE_DECL()
#define GEMM_TRACE_MARK(V)
#define GEMM_TRACE_END()
#endif

template <bool F16>
VBX_DEV f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// acc[i][j]: lane (m = lane & 15, g = lane >> 4) holds C[row0 + (i>>1)*hstep + (i&1)*16 + m][col0 + j*16 + 4g + r], r = 0..3.
// gemm3: hstep = 128, row0 = tile row + wr*32 (a wave owns 32 rows of each 128-row half of the tile, gemm3_layout.hpp);
// gemm4: hstep = 32, row0 = tile row + wr*64 (64 consecutive rows).
typedef f32x4 Acc[4][8];
#define G3_ROW(row0, i, m) ((row0) + ((i) >> 1) * hstep + ((i) & 1) * 16 + (m))

VBX_DEV uint2 pack4_bf16(const f32x4& v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
VBX_DEV uint2 pack4_f16(const f32x4& v) { return make_uint2(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3])); }
VBX_DEV uint2 pack4_f16_sat(const f32x4& v) { return make_uint2(pack_f16x2_sat(v[0], v[1]), pack_f16x2_sat(v[2], v[3])); }
VBX_DEV f32x4 ld4(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  return (f32x4){t.x, t.y, t.z, t.w};
}
VBX_DEV void st4(float* p, const f32x4& v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

// ---- row-contiguous 16-bit stores through LDS --------------------------------------------------------------------------
// Stored straight from the accumulators a wave instruction touches 16 rows x 32 bytes: 16 cache lines for 512 bytes.  Measured
// (tools/probes/store_pattern.hip, a 33 MB bf16 output from 256 workgroups): 12.2 us that way, 8.0 us with 64-byte runs, 6.8 us
// with whole rows -- and in the GEMMs (tools/native/gemm_trace.cpp) the register epilogue of a 256 x 256 tile took 8.6 us (plain
// bf16) to 17 us (to_qkv with all copies) against a 14 us k-loop.  So a wave transposes through LDS: its 64 x 128 block goes in
// two 32-row passes through a wave-private region of the (now dead) operand ring, is read back row-major and leaves as 16 bytes
// per lane, 4 rows x 256 B (or 8 rows x 128 B) per instruction.  No workgroup barrier: a wave only touches its own region, and LDS
// operations of one wave execute in order.  The kernel guarantees that every wave is done with the ring before the first put().
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int EPI_STAGE_BYTES = 16384;  // per wave

template <int NJ>  // 16-column blocks per row: 8 (the wave's 128 columns) or 4 (64 columns: the GEGLU output)
struct RowStage {
  static constexpr int STRIDE = NJ * 32, CPR = 2 * NJ, RPI = 64 / CPR, ITS = 32 / RPI, BYTES = 32 * STRIDE;
  // the index functions live in epi_stage_layout.hpp (checked on the host: tests/native/epi_stage_check.cpp)
  static VBX_DEV void put(unsigned buf, int il, int j, int lane, u32x2 v) {
    const unsigned a = buf + epst::put_byte(NJ, il, j, lane);
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
  }
  static VBX_DEV int row_of(int it, int lane) { return epst::get_row(NJ, it, lane); }
  static VBX_DEV int chunk_of(int lane) { return epst::get_chunk(NJ, lane); }
  static VBX_DEV u32x4 get(unsigned buf, int it, int lane) {
    const unsigned a = buf + epst::get_byte(NJ, it, lane);
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
  }
  // read the staged 32 rows back and store them: addr(row in pass, chunk) -> destination of the 8 values or nullptr
  template <class Addr>
  static VBX_DEV void flush(unsigned buf, int lane, const Addr& addr) {
    u32x4 v[ITS];
#pragma unroll
    for (int it = 0; it < ITS; it++) v[it] = get(buf, it, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < ITS; it++) {
      u16* p = addr(row_of(it, lane), chunk_of(lane));
      if (p) *reinterpret_cast<uint4*>(p) = make_uint4(v[it][0], v[it][1], v[it][2], v[it][3]);
    }
  }
};
VBX_DEV u32x2 as_u32x2(const uint2& v) { return (u32x2){v.x, v.y}; }

struct Epi3BF16 {
  u16* C; long ldc; const float* bias;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep, unsigned stage) const {
    const int g = lane >> 4;
    f32x4 bv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int gc = col0 + j * 16 + 4 * g;
      bv[j] = (bias && gc < N) ? ld4(bias + gc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ih = 0; ih < 2; ih++) {
      const unsigned buf = stage + ih * RowStage<8>::BYTES;
#pragma unroll
      for (int il = 0; il < 2; il++)
#pragma unroll
        for (int j = 0; j < 8; j++) RowStage<8>::put(buf, il, j, lane, as_u32x2(pack4_bf16(acc[2 * ih + il][j] + bv[j])));
      const int rbase = row0 + ih * hstep;
      RowStage<8>::flush(buf, lane, [&](int rr, int c) -> u16* {
        const int gr = rbase + rr, gc = col0 + c * 8;
        return (gr < M && gc < N) ? C + (long)gr * ldc + gc : nullptr;
      });
    }
  }
};

// The dgrad of Attention.to_out (dO = dX . W_out, bf16) with the attention backward's delta = rowsum_d(dO * O) per (token, head)
// taken on the way out (round 5): the row stage hands every lane 8 consecutive bf16 dO values of one (row, head) -- the lane loads the
// matching 16 bytes of the forward output O (fp16, the same [M, ldc] token-major layout), multiplies, and the 8 lanes of a head's 64
// columns add up by three lane exchanges.  Replaces attn_delta_kernel's separate 34 MB pass (8.9 us per layer) with one 16-byte load
// per store of this epilogue.  The products use the bf16-ROUNDED dO, as the stand-alone kernel (and the dP of the backward) do.
struct Epi3BF16Delta {
  u16* C; long ldc; const u16* O; float* delta; int H, Np;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep, unsigned stage) const {
    typedef RowStage<8> RS;
#pragma unroll
    for (int ih = 0; ih < 2; ih++) {
      const unsigned buf = stage + ih * RS::BYTES;
#pragma unroll
      for (int il = 0; il < 2; il++)
#pragma unroll
        for (int j = 0; j < 8; j++) RS::put(buf, il, j, lane, as_u32x2(pack4_bf16(acc[2 * ih + il][j])));
      const int rbase = row0 + ih * hstep;
      const int c = RS::chunk_of(lane);
      const int gc = col0 + c * 8;
#pragma unroll
      for (int half = 0; half < 2; half++) {  // two groups of ITS / 2 read-backs: bounds the live registers (4 x (dO + O) chunks)
        constexpr int NI = RS::ITS / 2;
        u32x4 v[NI];
        uint4 ov[NI];
#pragma unroll
        for (int u = 0; u < NI; u++) v[u] = RS::get(buf, half * NI + u, lane);
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int gr = rbase + RS::row_of(half * NI + u, lane);
          const bool ok = gr < M && gc < N;
          ov[u] = ok ? *reinterpret_cast<const uint4*>(O + (long)gr * ldc + gc) : make_uint4(0u, 0u, 0u, 0u);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NI; u++) {
          const int gr = rbase + RS::row_of(half * NI + u, lane);
          const bool ok = gr < M && gc < N;
          if (ok) *reinterpret_cast<uint4*>(C + (long)gr * ldc + gc) = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
          const unsigned ow[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w};
          float sdot = 0.f;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sdot += f16_to_f32((u16)(ow[e] & 0xffff)) * bf16_to_f32((u16)(v[u][e] & 0xffff));
            sdot += f16_to_f32((u16)(ow[e] >> 16)) * bf16_to_f32((u16)(v[u][e] >> 16));
          }
          sdot += __shfl_xor(sdot, 1, 64);
          sdot += __shfl_xor(sdot, 2, 64);
          sdot += __shfl_xor(sdot, 4, 64);
          if (ok && (c & 7) == 0) {
            const int b = gr / Np, n = gr - b * Np;
            delta[((long)b * H + (gc >> 6)) * Np + n] = sdot;
          }
        }
      }
    }
  }
};

struct Epi3F32 {
  float* C; long ldc; const float* bias; const float* resid; u16* C2;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep, unsigned) const {
    const int m = lane & 15, g = lane >> 4;
    f32x4 bv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int gc = col0 + j * 16 + 4 * g;
      bv[j] = (bias && gc < N) ? ld4(bias + gc) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gr = G3_ROW(row0, i, m);
      if (gr >= M) continue;
      f32x4 rv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {  // all residual loads of the row before the first use
        const int gc = col0 + j * 16 + 4 * g;
        rv[j] = (resid && gc < N) ? ld4(resid + (long)gr * ldc + gc) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int gc = col0 + j * 16 + 4 * g;
        if (gc >= N) continue;
        const f32x4 v = acc[i][j] + bv[j] + rv[j];
        st4(C + (long)gr * ldc + gc, v);
        if (C2) *reinterpret_cast<uint2*>(C2 + (long)gr * ldc + gc) = pack4_bf16(v);
      }
    }
  }
};

struct Epi3SplitK {
  float* C;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int split, int M, int N, int hstep, unsigned) const {
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gr = G3_ROW(row0, i, m);
      if (gr >= M) continue;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int gc = col0 + j * 16 + 4 * g;
        if (gc < N) st4(C + ((long)split * M + gr) * N + gc, acc[i][j]);
      }
    }
  }
};

// FeedForward[0] + GEGLU (voicebox_pytorch.py:338-340,345).  Packed weight rows: every 128-column block holds 64 "x" columns
// followed by their 64 "gate" columns; a wave's 128 columns are exactly one block, so x (j = 0..3) and gate (j + 4) of the same
// hidden unit sit in the same lane.
struct Epi3GEGLU {
  u16* G; long ldg; const float* bias; u16* H1; long ldh; u16* Gb; int g_f16;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep, unsigned stage) const {
    const int g = lane >> 4;
    if (col0 >= N) return;  // N is a multiple of 128: a wave's block is entirely in or out
    f32x4 bv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) bv[j] = ld4(bias + col0 + j * 16 + 4 * g);
    const unsigned bG = stage, bGb = stage + RowStage<4>::BYTES, bH = stage + 2 * RowStage<4>::BYTES;  // 4 + 4 + 8 KiB
#pragma unroll
    for (int ih = 0; ih < 2; ih++) {
#pragma unroll
      for (int il = 0; il < 2; il++) {
        const int i = 2 * ih + il;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const f32x4 x = acc[i][j] + bv[j];
          const f32x4 gt = acc[i][j + 4] + bv[j + 4];
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = gelu_erf(gt[r]) * x[r];
          RowStage<4>::put(bG, il, j, lane, as_u32x2(g_f16 ? pack4_f16_sat(o) : pack4_bf16(o)));
          if (Gb) RowStage<4>::put(bGb, il, j, lane, as_u32x2(pack4_bf16(o)));
        }
        if (H1) {
#pragma unroll
          for (int j = 0; j < 8; j++) RowStage<8>::put(bH, il, j, lane, as_u32x2(pack4_bf16(acc[i][j] + bv[j])));
        }
      }
      const int rbase = row0 + ih * hstep;
      const int gcol = (col0 >> 1);
      RowStage<4>::flush(bG, lane, [&](int rr, int c) -> u16* {
        const int gr = rbase + rr;
        return gr < M ? G + (long)gr * ldg + gcol + c * 8 : nullptr;
      });
      if (Gb)
        RowStage<4>::flush(bGb, lane, [&](int rr, int c) -> u16* {
          const int gr = rbase + rr;
          return gr < M ? Gb + (long)gr * ldg + gcol + c * 8 : nullptr;
        });
      if (H1)
        RowStage<8>::flush(bH, lane, [&](int rr, int c) -> u16* {
          const int gr = rbase + rr;
          return gr < M ? H1 + (long)gr * ldh + col0 + c * 8 : nullptr;
        });
    }
  }
};

// to_qkv + MultiheadRMSNorm + rotary, written head-major (voicebox_pytorch.py:320-328).  A wave's 128 columns are two heads
// (j >> 2); a head's 64 columns of one row sit in the 4 lanes that share (lane & 15): the sum of squares is 16 in-lane terms
// and two cross-lane adds, rotate_half pairs d and d + 32 are blocks j and j + 2 of the same lane.  The results go through the
// row stage: a (token, head) row is 128 contiguous bytes, 8 of them per store instruction.
struct Epi3QKV {
  int Np, H;
  float qk_scale;
  const float* qg; const float* kg; const float* rc; const float* rs;
  u16* q16; u16* k16; u16* qb; u16* kb; u16* v; float* qrn; float* krn; u16* v16;
  float qps;  // q16 = q-hat * qps (the attention kernels' contract, include/vbx.h); qb, k16, kb unscaled
  // (batch, token) of global row gr, given those of the pass's first row (a pass is 32 consecutive rows)
  VBX_DEV void split_row(int gr, int rbase, int b0, int n0, int& b, int& n) const {
    if (Np >= 32) {
      b = b0; n = n0 + (gr - rbase);
      if (n >= Np) { n -= Np; b++; }
    } else {
      b = gr / Np; n = gr - b * Np;
    }
  }
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep, unsigned stage) const {
    const int m = lane & 15, g = lane >> 4;
    if (col0 >= N) return;
    const int I = H * 64;
    const int which = col0 / I;
    const int hbase = (col0 - which * I) >> 6;
    u16* dst16 = which == 0 ? q16 : (which == 1 ? k16 : v16);
    u16* dstb = which == 0 ? qb : (which == 1 ? kb : v);
    const unsigned b16 = stage, bbf = stage + RowStage<8>::BYTES;
#pragma unroll
    for (int ih = 0; ih < 2; ih++) {
      const int rbase = row0 + ih * hstep;
      const int rb = __builtin_amdgcn_readfirstlane(min(rbase, M - 1));
      const int b0 = rb / Np, n0 = rb - b0 * Np;
#pragma unroll
      for (int il = 0; il < 2; il++) {
        const int i = 2 * ih + il;
        if (which == 2) {  // v: plain head split
#pragma unroll
          for (int j = 0; j < 8; j++) {
            if (v16) RowStage<8>::put(b16, il, j, lane, as_u32x2(pack4_f16_sat(acc[i][j])));
            if (v) RowStage<8>::put(bbf, il, j, lane, as_u32x2(pack4_bf16(acc[i][j])));
          }
          continue;
        }
        const int gr = rbase + il * 16 + m;
        const bool valid = gr < M;
        int b, n;
        split_row(valid ? gr : rb, rb, b0, n0, b, n);
        // cross-lane sums are taken by every lane (also those of out-of-range rows: they hold finite zeros-products)
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
          const int head = hbase + hh;
          f32x4 t[4];
#pragma unroll
          for (int jj = 0; jj < 4; jj++) t[jj] = acc[i][hh * 4 + jj];
          float ss = 0.f;
#pragma unroll
          for (int jj = 0; jj < 4; jj++)
#pragma unroll
            for (int r = 0; r < 4; r++) ss += t[jj][r] * t[jj][r];
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          const float rinv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
          if (qk_scale > 0.f) {
            const float* gam = (which == 0 ? qg : kg) + head * 64 + 4 * g;
            const float rs_ = rinv * qk_scale;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
              const f32x4 gv = ld4(gam + jj * 16);
#pragma unroll
              for (int r = 0; r < 4; r++) t[jj][r] = t[jj][r] * rs_ * gv[r];
            }
          }
          // rotate_half (voicebox_pytorch.py:193-199): out[d] = t[d] cos - t[d+32] sin (d < 32), out[d+32] = t[d+32] cos + t[d] sin
          f32x4 o4[4];
#pragma unroll
          for (int jj = 0; jj < 2; jj++) {
            const f32x4 c4 = ld4(rc + (long)n * 32 + jj * 16 + 4 * g);
            const f32x4 s4 = ld4(rs + (long)n * 32 + jj * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; r++) {
              o4[jj][r] = t[jj][r] * c4[r] - t[jj + 2][r] * s4[r];
              o4[jj + 2][r] = t[jj + 2][r] * c4[r] + t[jj][r] * s4[r];
            }
          }
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            if (dstb) RowStage<8>::put(bbf, il, hh * 4 + jj, lane, as_u32x2(pack4_bf16(o4[jj])));
            if (which == 0) o4[jj] *= qps;
            RowStage<8>::put(b16, il, hh * 4 + jj, lane, as_u32x2(pack4_f16(o4[jj])));
          }
          float* rn = (which == 0 ? qrn : krn);
          if (valid && rn && g == 0) rn[((long)b * H + head) * Np + n] = rinv;
        }
      }
      auto addr_of = [&](u16* dst, int rr, int c) -> u16* {
        const int gr = rbase + rr;
        if (gr >= M) return nullptr;
        int b, n;
        split_row(gr, rb, b0, n0, b, n);
        return dst + (((long)b * H + hbase + (c >> 3)) * Np + n) * 64 + (c & 7) * 8;
      };
      if (dst16) RowStage<8>::flush(b16, lane, [&](int rr, int c) -> u16* { return addr_of(dst16, rr, c); });
      if (dstb) RowStage<8>::flush(bbf, lane, [&](int rr, int c) -> u16* { return addr_of(dstb, rr, c); });
    }
  }
};


}  // namespace gepi
