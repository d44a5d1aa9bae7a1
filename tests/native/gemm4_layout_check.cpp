// Host emulation of gemm4.hip's data movement (no GPU), as gemm3_layout_check.cpp: LDS-DMA slots -> LDS image -> fragment reads
// -> transposed MFMA accumulators -> C, through csrc/gemm4_layout.hpp, against a plain GEMM; bank checks; immediate identities.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include "../../voicebox-pytorch_amd/csrc/gemm4_layout.hpp"
using namespace g4;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

struct Operand {
  int mode, outer_n, K;
  std::vector<int16_t> v;
  int16_t at(int o, int k) const { return mode == 0 ? v[(size_t)o * K + k] : v[(size_t)k * outer_n + o]; }
};

// Stage4<MODE, IS_A>::issue (gemm4.hip): piece q of thread tid -> (outer in tile, k in step) and LDS slot
static void fill_operand(const Operand& X, bool is_a, int o0, int k0, uint8_t* op) {
  const int nd = is_a ? A_DMAS : B_DMAS;
  for (int tid = 0; tid < THREADS; tid++)
    for (int q = 0; q < nd; q++) {
      int o, k;
      if (X.mode == 0) kc_slot(tid, o, k); else ks_slot(tid, o, k);
      int dout, dk;
      if (X.mode == 0) { dout = 64 * q; dk = 0; } else { dout = 128 * (q >> 1); dk = 16 * (q & 1); }
      const int outer = o + dout, kk = k + dk;
      const int s = q * THREADS + tid;
      // the slot's own formula (per image for KS) must agree
      int o2, k2;
      if (X.mode == 0) { kc_slot(s, o2, k2); }
      else { ks_slot(s & 511, o2, k2); o2 += 128 * (s >> 9); }
      CHECK(o2 == outer && k2 == kk, "piece mode %d is_a %d tid %d q %d: (%d,%d) vs (%d,%d)", X.mode, is_a, tid, q, outer, kk, o2, k2);
      int16_t e[8];
      for (int u = 0; u < 8; u++) {
        const int go = o0 + outer + (X.mode == 0 ? 0 : u), gk = k0 + kk + (X.mode == 0 ? u : 0);
        e[u] = (go < X.outer_n && gk < X.K) ? X.at(go, gk) : 0;
      }
      memcpy(op + s * 16, e, 16);
    }
}

static void read_frag(int mode, const uint8_t* op, int o_w, int F, int lane, int16_t out[8]) {
  const int o = o_w + F * 16;
  if (mode == 0) {
    // kernel: a[0] = kc_frag_byte(o_w, lane), immediate F*1024
    CHECK(kc_frag_byte(o, lane) == kc_frag_byte(o_w, lane) + F * 1024, "KC immediate o_w %d F %d lane %d", o_w, F, lane);
    memcpy(out, op + kc_frag_byte(o, lane), 16);
  } else {
    const int qtr = lane & ~15, a = lane & 15;
    for (int hi = 0; hi < 2; hi++) {
      CHECK(ks_frag_byte(o & 127, lane, hi) == ks_frag_byte(o & 127, lane, 0) + hi * 1024, "KS hi immediate");
      for (int j = 0; j < 4; j++) {
        const int supplier = qtr + 4 * j + (a >> 2);
        const uint8_t* p = op + (o >> 7) * KS_IMAGE + ks_frag_byte(o & 127, supplier, hi);
        int16_t e[4];
        memcpy(e, p, 8);
        out[hi * 4 + j] = e[a & 3];
      }
    }
  }
}

static void run(int ma, int mb, int M, int N, int K, int m0, int n0) {
  Operand A{ma, M, K, {}}, B{mb, N, K, {}};
  A.v.resize((size_t)M * K); B.v.resize((size_t)N * K);
  for (auto& x : A.v) x = (int16_t)(rand() % 7 - 3);
  for (auto& x : B.v) x = (int16_t)(rand() % 7 - 3);
  std::vector<long> C((size_t)BM * BN, 0);
  std::vector<uint8_t> lds(STAGE);
  for (int k0 = 0; k0 < K; k0 += BK) {
    fill_operand(A, true, m0, k0, lds.data());
    fill_operand(B, false, n0, k0, lds.data() + A_BYTES);
    for (int wave = 0; wave < 4; wave++) {
      const int wr = wave >> 1, wc = wave & 1;
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) {
          int16_t af[64][8], bf[64][8];
          for (int lane = 0; lane < 64; lane++) {
            read_frag(ma, lds.data(), wr * 64, i, lane, af[lane]);
            read_frag(mb, lds.data() + A_BYTES, wc * 128, j, lane, bf[lane]);
          }
          for (int np = 0; np < 16; np++)
            for (int mp = 0; mp < 16; mp++) {
              long s = 0;
              for (int g = 0; g < 4; g++)
                for (int u = 0; u < 8; u++) s += (long)bf[g * 16 + np][u] * af[g * 16 + mp][u];
              // epilogue (hstep = 32): row = wr*64 + (i>>1)*32 + (i&1)*16 + m = wr*64 + i*16 + m; col = wc*128 + j*16 + 4g + r = .. + np
              C[(size_t)(wr * 64 + i * 16 + mp) * BN + wc * 128 + j * 16 + np] += s;
            }
        }
    }
  }
  for (int r = 0; r < BM; r++)
    for (int c = 0; c < BN; c++) {
      long ref = 0;
      if (m0 + r < M && n0 + c < N)
        for (int k = 0; k < K; k++) ref += (long)A.at(m0 + r, k) * B.at(n0 + c, k);
      CHECK(C[(size_t)r * BN + c] == ref, "modes (%d,%d) M %d N %d K %d tile (%d,%d): C[%d][%d] = %ld, expected %ld", ma, mb, M, N, K, m0, n0, r, c,
            C[(size_t)r * BN + c], ref);
    }
}

static void banks() {
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  for (int o16 = 0; o16 < 256; o16 += 16) {
    for (int gi = 0; gi < 4; gi++) {
      int used[64] = {0};
      for (int x = 0; x < 16; x++) {
        const int b0 = (kc_frag_byte(o16, grp[gi][x]) / 4) % 64;
        for (int d = 0; d < 4; d++) used[(b0 + d) % 64]++;
      }
      for (int b = 0; b < 64; b++) CHECK(used[b] == 1, "KC bank conflict o16 %d group %d bank %d x%d", o16, gi, b, used[b]);
    }
    if (o16 < 128)
      for (int hi = 0; hi < 2; hi++)
        for (int half = 0; half < 2; half++) {
          int used[64] = {0};
          for (int l = half * 32; l < half * 32 + 32; l++) {
            const int b0 = (ks_frag_byte(o16, l, hi) / 4) % 64;
            used[b0]++; used[(b0 + 1) % 64]++;
          }
          for (int b = 0; b < 64; b++) CHECK(used[b] == 1, "KS bank conflict o16 %d hi %d half %d bank %d x%d", o16, hi, half, b, used[b]);
        }
  }
}

int main() {
  srand(2);
  banks();
  for (int ma = 0; ma < 2; ma++)
    for (int mb = 0; mb < 2; mb++) {
      run(ma, mb, 128, 256, 64, 0, 0);
      run(ma, mb, 384, 768, 32, 256, 512);
      run(ma, mb, 200, 296, 72, 128, 256);
    }
  printf(fails ? "gemm4 layout check: %d FAILURES\n" : "gemm4 layout check: ok\n", fails);
  return fails ? 1 : 0;
}
