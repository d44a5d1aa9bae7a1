// LDS image of the 128 x 256 x 32 GEMM tile of gemm4.hip (4 waves, 2 x 2, each 64 x 128; 3-stage ring, two workgroups per CU),
// as plain index functions shared by the kernel and the host emulation (tests/native/gemm4_layout_check.cpp).
// Same scheme as gemm.hip's 128 x 128 kernel (validated on hardware in round 1): the LDS-DMA destination is lane-linear, so
// layouts are produced by permuting the per-lane SOURCE address and applying the same permutation on the fragment read.
#pragma once
#ifdef __HIPCC__
#define G4_HD __host__ __device__ inline
#else
#define G4_HD inline
#endif

namespace g4 {

constexpr int BM = 128, BN = 256, BK = 32, NST = 3, THREADS = 256;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;  // 8 KiB, 16 KiB
constexpr int STAGE = A_BYTES + B_BYTES, LDS_BYTES = NST * STAGE;  // 24 KiB, 72 KiB
constexpr int A_DMAS = A_BYTES / (THREADS * 16), B_DMAS = B_BYTES / (THREADS * 16);  // 2, 4 LDS-DMA instructions per thread per stage

// ---- K-contiguous operand: [outer][32 k], 64 bytes per outer index; rows 2p, 2p+1 share a 128-byte line whose eight
// 16-byte slots are permuted by (p & 7): slot of chunk c (8 k values) of row r is (c + 4*(r & 1)) ^ (p & 7)
G4_HD int kc_byte(int o, int k) {
  const int p = o >> 1;
  return p * 128 + (((((k >> 3) & 3) + 4 * (o & 1)) ^ (p & 7)) << 4) + (k & 7) * 2;
}
G4_HD void kc_slot(int s, int& o, int& k) {  // DMA slot s (LDS byte s*16) -> first element
  const int p = s >> 3, x = (s & 7) ^ (p & 7);
  o = 2 * p + (x >> 2);
  k = (x & 3) * 8;
}
// ---- K-strided operand: images of [32 k][128 outer], 256 bytes per k row (B: two images, columns 0-127 and 128-255);
// 32-byte unit P of k row k sits in unit P ^ ks_f(k)
G4_HD int ks_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
G4_HD int ks_byte(int k, int o) { return k * 256 + ((((o >> 4) ^ ks_f(k)) & 7) << 5) + (o & 15) * 2; }
G4_HD void ks_slot(int s, int& o, int& k) {  // s in 0..511 of one image
  k = s >> 4;
  const int pc = s & 15;
  const int P = ((pc >> 1) ^ ks_f(k)) & 7;
  o = (2 * P + (pc & 1)) * 8;
}
constexpr int KS_IMAGE = 32 * 256;  // 8 KiB

// ---- fragment reads.  MFMA 16x16x32 operand: lane l holds outer index (l & 15), k = (l >> 4)*8 .. +8 of the 32-deep step.
G4_HD int kc_frag_byte(int o16, int lane) { return kc_byte(o16 + (lane & 15), (lane >> 4) * 8); }
// two ds_read_b64_tr_b16 (hi = k + 4); o16 is local to the 128-wide image
G4_HD int ks_frag_byte(int o16, int lane, int hi) {
  const int g = lane >> 4, a = lane & 15;
  return ks_byte(g * 8 + (a >> 2) + 4 * hi, o16 + 4 * (a & 3));
}

}  // namespace g4
