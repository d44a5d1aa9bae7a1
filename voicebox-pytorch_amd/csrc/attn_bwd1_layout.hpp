// Index functions of the one-pass attention backward (attn_bwd1.inc), shared by the kernel and the host emulation
// (tests/native/attn_bwd1_layout_check.cpp): the work queue's item map, the dS^T LDS image (transposing write, transposed fragment
// read, the XOR identities that let one base register serve eight addresses) and the fragment-order scratch of the running dq sums.
#pragma once
#ifdef __HIPCC__
#define B1_HD __host__ __device__ constexpr inline
#else
#define B1_HD constexpr inline
#endif

namespace b1 {

constexpr int TILE = 64 * 128;  // one [64 rows][64 x 16-bit] tile, 16-byte chunks XOR-swizzled: 8192 B

// 16-byte-chunk swizzle key of a tile row (attn.hip attn_swz): bijection of (row >> 1) mod 8; key(row + 16) == key(row)
B1_HD int swz(int row) {
  const int p = row >> 1;
  return ((p & 1) << 2) | ((p >> 1) & 3);
}
B1_HD int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ swz(row)) << 4); }

// ---- work queue of XCD x (heads bh = x + nx * i, i < nh): key-block-major, so that chain member kb of a head always sits BEHIND
// member kb - 1 in the queue it is pulled from
B1_HD int heads_of_xcd(int BH, int nx, int x) { return x < BH ? (BH - x + nx - 1) / nx : 0; }
B1_HD void queue_item(int j, int nh, int nx, int x, int& bh, int& kb) {
  kb = j / nh;
  bh = x + nx * (j - kb * nh);
}

// ---- dS^T buffer: two tiles [64 keys][64 q] bf16 (keys 0-63, 64-127 of the workgroup).  Wave w, lane l owns key row w*32 + (l&31);
// after block QB its registers 4g .. 4g+3 hold q = QB*32 + 8g + 4*(l>>5) + 0..3: one 8-byte write per g.
// Base (one register): the row's byte 0 of chunk 0 ^ key, plus the 8-byte half; write (QB, g) goes to base ^ ((QB*4 + g) << 4).
B1_HD int ds_write_base(int wave, int lane) {
  const int kr = (wave & 1) * 32 + (lane & 31);
  return (wave >> 1) * TILE + kr * 128 + (swz(kr) << 4) + 8 * (lane >> 5);
}
B1_HD int ds_write_off(int wave, int lane, int QB, int g) { return ds_write_base(wave, lane) ^ ((QB * 4 + g) << 4); }

// Transposed fragment read (ds_read_b64_tr_b16) of 16 rows [rbase, rbase+16) x 32 columns [c0, c0+32): the address THIS lane supplies
// for the rows +0..7 part (`plus8` = 0) and the rows +8..15 part (1).  Inside each 16-lane group, lane a then RECEIVES, for j = 0..3,
// element (a & 3) of the 8 bytes addressed by lane 4j + (a >> 2) (tests/test_ops_gpu.py::test_probe_tr16), i.e. column c0 + 16*(G&1) + a
// of rows rbase + 8*plus8 + 4*(G>>1) + j.
B1_HD int tr_addr(int lane, int c0, int plus8) {
  const int G = lane >> 4, a = lane & 15;
  const int row = 4 * (G >> 1) + (a >> 2) + 8 * plus8;
  const int col = c0 + (G & 1) * 16 + 4 * (a & 3);
  return tile_off(row, col >> 3) + (col & 7) * 2;
}
// k-step ks (16 keys) of the 128-key buffer: DS immediate added to the base above
B1_HD int kstep_imm(int ks) { return (ks >> 2) * TILE + (ks & 3) * 2048; }

// ---- running dq sums, "fragment order": [bh][tile][wave][r4][lane][4 floats]
B1_HD long acc_index(long bh, int ntiles, int qt, int wave, int r4, int lane, int j) {
  return (((bh * ntiles + qt) * 4 + wave) * 4 + r4) * 256 + lane * 4 + j;
}
// wave (qh = w >> 1, dh = w & 1), lane, accumulator register r  ->  (query row in the 64-row tile, dim)
B1_HD void acc_coord(int wave, int lane, int r, int& q, int& d) {
  q = (wave >> 1) * 32 + (lane & 31);
  d = (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

}  // namespace b1
