// LDS image of the 256 x 256 x 64 GEMM tile (gemm3.hip) as plain index functions, shared by the kernel and by the host-side
// emulation test (tests/native/gemm3_layout_check.cpp): the LDS-DMA destination is lane-linear, so where an element lands is
// decided by which SOURCE address a lane is given, and the fragment reads must apply the same permutation.  Getting one side
// wrong is silent garbage; the emulation replays DMA slots and fragment reads through these functions against a plain GEMM.
#pragma once
#ifdef __HIPCC__
#define G3_HD __host__ __device__ inline
#else
#define G3_HD inline
#endif

namespace g3 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int THREADS = 512;
// one k-tile buffer = four 16 KiB regions; a region is what ONE phase of the k-loop reads and what one phase re-stages
//   A-lo: tile rows [0,128)    A-hi: tile rows [128,256)     wave row wr (0..3) owns rows wr*32 + [0,32) of EACH half, so a
//         region is one contiguous 128-row block (full 256-byte runs when the operand is k-strided)
//   B-lo: cols wc*128 + [0,64) of the 2 wave columns  B-hi: cols wc*128 + [64,128)   (a wave's 128 columns stay contiguous:
//         the GEGLU / qk-norm epilogues pair columns inside one 128-column block)
constexpr int REGION = 16384, BUF = 4 * REGION, LDS_BYTES = 2 * BUF;
constexpr int OFF_ALO = 0, OFF_AHI = REGION, OFF_BLO = 2 * REGION, OFF_BHI = 3 * REGION;
// outer-index distance of a thread's second DMA piece (slot + 512, K-contiguous operands) and of the hi region
constexpr int A_QSTEP = 64, A_HISTEP = 128, B_QSTEP = 128, B_HISTEP = 64;
// accumulator block i (0..3) of wave row wr, lane row m (0..15) -> tile row
G3_HD int acc_row(int wr, int i, int m) { return (i >> 1) * 128 + wr * 32 + (i & 1) * 16 + m; }

// region-local outer index o (0..127) -> outer index inside the 256-wide tile
G3_HD int a_outer(int o, int hi) { return hi * 128 + o; }
G3_HD int b_outer(int o, int hi) { return (o >> 6) * 128 + hi * 64 + (o & 63); }

// ---- K-contiguous operand ("KC": the tile is [128 outer][64 k], 128 bytes per outer index)
// 16-byte chunk c (8 k values) of outer index o sits in slot c ^ (o & 7) of its 128-byte row: a ds_read_b128 of one MFMA
// fragment (16 consecutive outer indices x one chunk per 16-lane quarter) then touches 16 distinct 16-byte bank groups.
G3_HD int kc_byte(int o, int k) { return o * 128 + ((((k >> 3) ^ (o & 7)) & 7) << 4) + (k & 7) * 2; }
// DMA slot s (0..1023, LDS byte s*16) holds the 8 k values starting at k of outer index o
G3_HD void kc_slot(int s, int& o, int& k) {
  o = s >> 3;
  k = ((s & 7) ^ (o & 7)) * 8;
}

// ---- K-strided operand ("KS": the tile is [64 k][128 outer], 256 bytes per k row), read with ds_read_b64_tr_b16
// 32-byte unit P (16 outer indices) of k row k sits in unit P ^ ks_f(k): the 8 pieces of 32 bytes one half-wave of a
// transpose read touches (4 k rows x 2 quarter-waves) fall into 8 different bank groups.
G3_HD int ks_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
G3_HD int ks_byte(int k, int o) { return k * 256 + ((((o >> 4) ^ ks_f(k)) & 7) << 5) + (o & 15) * 2; }
// DMA slot s holds the 8 outer indices starting at o of k row k
G3_HD void ks_slot(int s, int& o, int& k) {
  k = s >> 4;
  const int pc = s & 15;
  const int P = ((pc >> 1) ^ ks_f(k)) & 7;
  o = (2 * P + (pc & 1)) * 8;
}

// ---- fragment reads.  MFMA 16x16x32 operand: lane l holds outer index (l & 15), k = kk*32 + (l >> 4)*8 .. +8.
// KC: one ds_read_b128 at this byte (o16 = region-local outer index of the fragment's first row, a multiple of 16)
G3_HD int kc_frag_byte(int o16, int kk, int lane) { return kc_byte(o16 + (lane & 15), kk * 32 + (lane >> 4) * 8); }
// KS: two ds_read_b64_tr_b16 (k .. k+3 and k+4 .. k+7).  Within a 16-lane quarter lane a receives element (a & 3) of the 8 bytes
// addressed by lanes 4j + (a >> 2), j = 0..3; so the lane ADDRESSES [k0 + (a >> 2)][o16 + 4*(a & 3)] and RECEIVES [k0 + j][o16 + a].
G3_HD int ks_frag_byte(int o16, int kk, int lane, int hi) {
  const int g = lane >> 4, a = lane & 15;
  return ks_byte(kk * 32 + g * 8 + (a >> 2) + 4 * hi, o16 + 4 * (a & 3));
}

}  // namespace g3
