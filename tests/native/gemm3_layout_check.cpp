// Host emulation of gemm3.hip's data movement (no GPU): LDS-DMA slots -> LDS image -> fragment reads -> transposed MFMA
// accumulators -> C, through the index functions of csrc/gemm3_layout.hpp, against a plain GEMM.  Checks what a wrong source
// permutation / read address / fragment-to-column map would silently break.  Build: g++ -O1 -std=c++17 (tests/test_host_cpu.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include "../../voicebox-pytorch_amd/csrc/gemm3_layout.hpp"
using namespace g3;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

// ---- the stride derivation Stage<MODE, IS_A>::init / issue use (gemm3.hip) -> (outer in tile, k in tile) of a piece
static void stage_piece(int mode, bool is_a, int tid, int q, int hi, int& outer, int& k) {
  int o, kk;
  if (mode == 0) kc_slot(tid, o, kk); else ks_slot(tid, o, kk);
  const int to = is_a ? a_outer(o, 0) : b_outer(o, 0);
  const int HI = is_a ? A_HISTEP : B_HISTEP, QS = is_a ? A_QSTEP : B_QSTEP;
  if (mode == 0) { outer = to + q * QS + hi * HI; k = kk; }
  else { outer = to + hi * HI; k = kk + q * 32; }
}

// elements are small integers stored as int16 (exact arithmetic)
struct Operand {
  int mode, outer_n, K;  // mode 0: [outer][K]; mode 1: [K][outer]
  std::vector<int16_t> v;
  int16_t at(int o, int k) const { return mode == 0 ? v[(size_t)o * K + k] : v[(size_t)k * outer_n + o]; }
};

static void fill_region(const Operand& X, bool is_a, int hi, int o0, int k0, uint8_t* region) {
  for (int tid = 0; tid < THREADS; tid++)
    for (int q = 0; q < 2; q++) {
      int outer, k;
      stage_piece(X.mode, is_a, tid, q, hi, outer, k);
      // direct slot formula of the layout header must agree with the stride derivation
      int o2, k2;
      const int s = q * THREADS + tid;
      if (X.mode == 0) kc_slot(s, o2, k2); else ks_slot(s, o2, k2);
      const int outer2 = is_a ? a_outer(o2, hi) : b_outer(o2, hi);
      CHECK(outer == outer2 && k == k2, "stride derivation mode %d is_a %d tid %d q %d hi %d: (%d,%d) vs (%d,%d)", X.mode, is_a, tid, q, hi, outer, k, outer2, k2);
      int16_t e[8];
      for (int u = 0; u < 8; u++) {
        const int go = o0 + outer + (X.mode == 0 ? 0 : u), gk = k0 + k + (X.mode == 0 ? u : 0);
        e[u] = (go < X.outer_n && gk < X.K) ? X.at(go, gk) : 0;
      }
      memcpy(region + s * 16, e, 16);  // LDS-DMA: wave base + lane * 16, wave base = (q*512 + wave*64)*16
    }
}

// fragment of region-local fragment index F (16 outer indices), k half kk, for `lane`: 8 values k = kk*32 + (lane>>4)*8 + u
static void read_frag(int mode, bool is_a, const uint8_t* region, int wq, int F, int kk, int lane, int16_t out[8]) {
  const int o_w = is_a ? wq * 32 : wq * 64;
  if (mode == 0) {
    memcpy(out, region + kc_frag_byte(o_w + F * 16, kk, lane), 16);
  } else {
    const int qtr = lane & ~15, a = lane & 15;
    for (int hi = 0; hi < 2; hi++)
      for (int j = 0; j < 4; j++) {
        const int supplier = qtr + 4 * j + (a >> 2);
        const uint8_t* p = region + ks_frag_byte(o_w + F * 16, kk, supplier, hi);
        int16_t e[4];
        memcpy(e, p, 8);
        out[hi * 4 + j] = e[a & 3];
      }
  }
}

static void run(int ma, int mb, int M, int N, int K, int m0, int n0) {
  Operand A{ma, M, K, {}}, B{mb, N, K, {}};
  A.v.resize((size_t)M * K); B.v.resize((size_t)N * K);
  for (auto& x : A.v) x = (int16_t)(rand() % 7 - 3);
  for (auto& x : B.v) x = (int16_t)(rand() % 7 - 3);
  std::vector<long> C((size_t)BM * BN, 0);
  std::vector<uint8_t> lds(BUF);
  for (int k0 = 0; k0 < K; k0 += BK) {
    fill_region(A, true, 0, m0, k0, lds.data() + OFF_ALO);
    fill_region(A, true, 1, m0, k0, lds.data() + OFF_AHI);
    fill_region(B, false, 0, n0, k0, lds.data() + OFF_BLO);
    fill_region(B, false, 1, n0, k0, lds.data() + OFF_BHI);
    // pairwise MFMA emulation per wave / fragment pair
    for (int wave = 0; wave < 8; wave++) {
      const int wr = wave >> 1, wc = wave & 1;
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++)
          for (int kk = 0; kk < 2; kk++) {
            int16_t af[64][8], bf[64][8];
            for (int lane = 0; lane < 64; lane++) {
              read_frag(ma, true, lds.data() + (i < 2 ? OFF_ALO : OFF_AHI), wr, i & 1, kk, lane, af[lane]);
              read_frag(mb, false, lds.data() + (j < 4 ? OFF_BLO : OFF_BHI), wc, j & 3, kk, lane, bf[lane]);
            }
            // D[n'][m'] = sum over k-groups g (lanes g*16 + n' of X, g*16 + m' of Y) and u
            for (int np = 0; np < 16; np++)
              for (int mp = 0; mp < 16; mp++) {
                long s = 0;
                for (int g = 0; g < 4; g++)
                  for (int u = 0; u < 8; u++) s += (long)bf[g * 16 + np][u] * af[g * 16 + mp][u];
                // accumulator register r of lane (m = mp, g = np >> 2) is D[np][mp], np = 4g + r; the epilogue maps it to
                // C[acc_row(wr, i, m)][wc*128 + j*16 + 4g + r]
                const int row = acc_row(wr, i, mp), col = wc * 128 + j * 16 + np;
                C[(size_t)row * BN + col] += s;
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

// bank check: every ds_read_b128 16-lane group and every transpose-read half wave touches each 4-byte bank at most once
static void banks() {
  static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                 {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  for (int o16 = 0; o16 < 128; o16 += 16)
    for (int kk = 0; kk < 2; kk++) {
      for (int gi = 0; gi < 4; gi++) {
        int used[64] = {0};
        for (int x = 0; x < 16; x++) {
          const int b0 = (kc_frag_byte(o16, kk, grp[gi][x]) / 4) % 64;
          for (int d = 0; d < 4; d++) used[(b0 + d) % 64]++;
        }
        for (int b = 0; b < 64; b++) CHECK(used[b] == 1, "KC bank conflict o16 %d kk %d group %d bank %d x%d", o16, kk, gi, b, used[b]);
      }
      for (int hi = 0; hi < 2; hi++)
        for (int half = 0; half < 2; half++) {
          int used[64] = {0};
          for (int l = half * 32; l < half * 32 + 32; l++) {
            const int b0 = (ks_frag_byte(o16, kk, l, hi) / 4) % 64;
            used[b0]++; used[(b0 + 1) % 64]++;
          }
          for (int b = 0; b < 64; b++) CHECK(used[b] == 1, "KS bank conflict o16 %d kk %d hi %d half %d bank %d x%d", o16, kk, hi, half, b, used[b]);
        }
    }
}

// the kernel folds fragment / k-half / hi offsets into DS immediates on top of one lane address (Frag::init, read_frag):
// check those identities for every wave position
static void immediates() {
  for (int lane = 0; lane < 64; lane++)
    for (int kk = 0; kk < 2; kk++)
      for (int o_w = 0; o_w < 128; o_w += 32)
        for (int F = 0; o_w + F * 16 < 128 && F < 4; F++) {
          CHECK(kc_frag_byte(o_w + F * 16, kk, lane) == kc_frag_byte(o_w, kk, lane) + F * 2048, "KC immediate lane %d kk %d o_w %d F %d", lane, kk, o_w, F);
          for (int hi = 0; hi < 2; hi++)
            CHECK(ks_frag_byte(o_w + F * 16, kk, lane, hi) == ks_frag_byte(o_w + F * 16, 0, lane, 0) + kk * 8192 + hi * 1024,
                  "KS immediate lane %d kk %d o_w %d F %d hi %d", lane, kk, o_w, F, hi);
        }
}

int main() {
  srand(1);
  banks();
  immediates();
  for (int ma = 0; ma < 2; ma++)
    for (int mb = 0; mb < 2; mb++) {
      run(ma, mb, 256, 256, 128, 0, 0);
      run(ma, mb, 512, 768, 64, 256, 512);
      run(ma, mb, 328, 296, 72, 256, 256);  // ragged M, N, K (multiples of 8)
    }
  printf(fails ? "gemm3 layout check: %d FAILURES\n" : "gemm3 layout check: ok\n", fails);
  return fails ? 1 : 0;
}
