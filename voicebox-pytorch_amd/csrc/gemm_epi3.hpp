// Register epilogues shared by the transposed-accumulator GEMM tiles (gemm3.hip: 256 x 256, gemm4.hip: 128 x 256).
#pragma once
#include "common.hpp"

namespace gepi {

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

struct Epi3BF16 {
  u16* C; long ldc; const float* bias;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep) const {
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
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int gc = col0 + j * 16 + 4 * g;
        if (gc < N) *reinterpret_cast<uint2*>(C + (long)gr * ldc + gc) = pack4_bf16(acc[i][j] + bv[j]);
      }
    }
  }
};

struct Epi3F32 {
  float* C; long ldc; const float* bias; const float* resid; u16* C2;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep) const {
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
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int split, int M, int N, int hstep) const {
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
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep) const {
    const int m = lane & 15, g = lane >> 4;
    if (col0 >= N) return;  // N is a multiple of 128: a wave's block is entirely in or out
    f32x4 bv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) bv[j] = ld4(bias + col0 + j * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gr = G3_ROW(row0, i, m);
      if (gr >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 x = acc[i][j] + bv[j];
        const f32x4 gt = acc[i][j + 4] + bv[j + 4];
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = gelu_erf(gt[r]) * x[r];
        const long go = (long)gr * ldg + (col0 >> 1) + j * 16 + 4 * g;
        *reinterpret_cast<uint2*>(G + go) = g_f16 ? pack4_f16_sat(o) : pack4_bf16(o);
        if (Gb) *reinterpret_cast<uint2*>(Gb + go) = pack4_bf16(o);
      }
      if (H1) {
#pragma unroll
        for (int j = 0; j < 8; j++)
          *reinterpret_cast<uint2*>(H1 + (long)gr * ldh + col0 + j * 16 + 4 * g) = pack4_bf16(acc[i][j] + bv[j]);
      }
    }
  }
};

// to_qkv + MultiheadRMSNorm + rotary, written head-major (voicebox_pytorch.py:320-328).  A wave's 128 columns are two heads
// (j >> 2); a head's 64 columns of one row sit in the 4 lanes that share (lane & 15): the sum of squares is 16 in-lane terms
// and two cross-lane adds, rotate_half pairs d and d + 32 are blocks j and j + 2 of the same lane.
struct Epi3QKV {
  int Np, H;
  float qk_scale;
  const float* qg; const float* kg; const float* rc; const float* rs;
  u16* q16; u16* k16; u16* qb; u16* kb; u16* v; float* qrn; float* krn; u16* v16;
  VBX_DEV void operator()(const Acc& acc, int row0, int col0, int lane, int, int M, int N, int hstep) const {
    const int m = lane & 15, g = lane >> 4;
    if (col0 >= N) return;
    const int I = H * 64;
    const int which = col0 / I;
    const int hbase = (col0 - which * I) >> 6;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int gr = G3_ROW(row0, i, m);
      const bool valid = gr < M;
      const int grc = valid ? gr : (M - 1);
      const int b = grc / Np, n = grc - b * Np;
      if (which == 2) {  // v: plain head split
        if (!valid) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const long o = (((long)b * H + hbase + (j >> 2)) * Np + n) * 64 + (j & 3) * 16 + 4 * g;
          if (v) *reinterpret_cast<uint2*>(v + o) = pack4_bf16(acc[i][j]);
          if (v16) *reinterpret_cast<uint2*>(v16 + o) = pack4_f16_sat(acc[i][j]);
        }
        continue;
      }
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
        if (valid) {
          const long ob = (((long)b * H + head) * Np + n) * 64 + 4 * g;
          u16* dst = (which == 0 ? q16 : k16);
          u16* bcopy = (which == 0 ? qb : kb);
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            *reinterpret_cast<uint2*>(dst + ob + jj * 16) = pack4_f16(o4[jj]);
            if (bcopy) *reinterpret_cast<uint2*>(bcopy + ob + jj * 16) = pack4_bf16(o4[jj]);
          }
          float* rn = (which == 0 ? qrn : krn);
          if (rn && g == 0) rn[((long)b * H + head) * Np + n] = rinv;
        }
      }
    }
  }
};


}  // namespace gepi
