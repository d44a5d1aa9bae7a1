// Host emulation (no GPU) of the index logic of the one-pass attention backward through csrc/attn_bwd1_layout.hpp:
//  1. queue: every (head, key block) of every XCD appears exactly once, and a chain member's predecessor precedes it in its queue;
//  2. dS^T image: the four waves' transposing 8-byte writes fill the buffer exactly once, and the transposed fragment reads
//     (ds_read_b64_tr_b16 semantics as probed on hardware) hand wave (qh, dh) lane l, slot s of k-step ks the element
//     dS[q = qh*32 + (l&31)][key = 16*ks + (s&3) + 8*(s>>2) + 4*(l>>5)] -- the order the Kb^T fragments use;
//  3. the XOR identities the kernel uses to derive addresses from one base register, with all DS immediates in range;
//  4. bank census of the writes (ds_write_b64: contiguous 16-lane groups, 32 banks) and of the transposed reads (32-lane halves,
//     64 banks);
//  5. fragment-order scratch: a bijection onto [0, 4096) per tile; accumulator coordinates cover the 64 x 64 tile once.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#include "../../voicebox-pytorch_amd/csrc/attn_bwd1_layout.hpp"
using namespace b1;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } fails++; } } while (0)

int main() {
  // ---- 1. queue
  for (int BH : {1, 2, 6, 8, 9, 15, 40, 128})
    for (int nx : {1, 2, 4, 8})
      for (int n_kb : {1, 2, 9}) {
        std::set<std::pair<int, int>> seen;
        for (int x = 0; x < nx; x++) {
          const int nh = heads_of_xcd(BH, nx, x);
          std::vector<std::pair<int, int>> order;
          for (int j = 0; j < nh * n_kb; j++) {
            int bh, kb;
            queue_item(j, nh, nx, x, bh, kb);
            CHECK(bh >= 0 && bh < BH && bh % nx == x && kb >= 0 && kb < n_kb, "item (%d,%d) BH %d nx %d x %d", bh, kb, BH, nx, x);
            CHECK(seen.insert({bh, kb}).second, "duplicate item (%d,%d)", bh, kb);
            if (kb > 0) {
              bool found = false;
              for (auto& o : order) found |= (o.first == bh && o.second == kb - 1);
              CHECK(found, "predecessor of (%d,%d) not ahead in the queue", bh, kb);
            }
            order.push_back({bh, kb});
          }
        }
        CHECK((int)seen.size() == BH * n_kb, "BH %d nx %d n_kb %d: %d items", BH, nx, n_kb, (int)seen.size());
      }

  // ---- 2. dS^T image
  std::vector<int32_t> lds(2 * TILE / 2, -1);  // one id per 16-bit element: key * 64 + q
  for (int w = 0; w < 4; w++)
    for (int l = 0; l < 64; l++)
      for (int QB = 0; QB < 2; QB++)
        for (int g = 0; g < 4; g++) {
          const int off = ds_write_off(w, l, QB, g);
          CHECK(off >= 0 && off + 8 <= 2 * TILE && off % 8 == 0, "write offset %d", off);
          for (int e = 0; e < 4; e++) {
            const int key = w * 32 + (l & 31), q = QB * 32 + 8 * g + 4 * (l >> 5) + e;
            CHECK(lds[off / 2 + e] == -1, "element written twice at %d", off / 2 + e);
            lds[off / 2 + e] = key * 64 + q;
          }
        }
  for (size_t i = 0; i < lds.size(); i++) CHECK(lds[i] >= 0, "hole at %zu", i);
  for (int w = 0; w < 4; w++)
    for (int ks = 0; ks < 8; ks++) {
      CHECK(kstep_imm(ks) >= 0 && kstep_imm(ks) < 65536, "immediate");
      for (int l = 0; l < 64; l++) {
        int got[8];
        for (int part = 0; part < 2; part++)
          for (int j = 0; j < 4; j++) {
            const int grp = l >> 4, a = l & 15;
            const int src_lane = grp * 16 + 4 * j + (a >> 2);
            const int addr = tr_addr(src_lane, (w >> 1) * 32, part) + kstep_imm(ks) + 2 * (a & 3);
            CHECK(addr % 2 == 0 && addr >= 0 && addr < 2 * TILE, "read address %d", addr);
            CHECK((tr_addr(src_lane, (w >> 1) * 32, part) & 7) == 0, "tr read address not 8-byte aligned");
            got[part * 4 + j] = lds[addr / 2];
          }
        for (int s = 0; s < 8; s++) {
          const int key = 16 * ks + (s & 3) + 8 * (s >> 2) + 4 * (l >> 5), q = (w >> 1) * 32 + (l & 31);
          CHECK(got[s] == key * 64 + q, "wave %d ks %d lane %d slot %d: got (%d,%d) want (%d,%d)", w, ks, l, s, got[s] / 64, got[s] % 64, key, q);
        }
      }
    }

  // ---- 3. XOR identities (row fragments: chunk 2t + hi; transposed fragments: d-half db; dS writes: chunk c)
  for (int l = 0; l < 64; l++) {
    const int row = l & 31, hi = l >> 5;
    for (int t = 0; t < 4; t++) CHECK(tile_off(row, 2 * t + hi) == (tile_off(row, hi) ^ (t << 5)), "row fragment identity lane %d t %d", l, t);
    for (int p8 = 0; p8 < 2; p8++) CHECK(tr_addr(l, 32, p8) == (tr_addr(l, 0, p8) ^ 64), "transposed fragment identity lane %d", l);
    for (int r16 = 0; r16 < 4; r16++)  // +16 rows keep the swizzle key: a plain immediate
      CHECK(tile_off(row + 16 * r16 < 64 ? row + 16 * r16 : row, 3) - tile_off(row, 3) == (row + 16 * r16 < 64 ? 16 * r16 * 128 : 0), "row + 16 immediate");
  }

  // ---- 4. bank census
  int worst_w = 0, worst_r = 0;
  for (int w = 0; w < 4; w++)
    for (int QB = 0; QB < 2; QB++)
      for (int g = 0; g < 4; g++)
        for (int grp = 0; grp < 4; grp++) {  // ds_write_b64: four contiguous 16-lane groups, bank = (addr / 4) % 32
          int cnt[32] = {0};
          for (int a = 0; a < 16; a++) {
            const int off = ds_write_off(w, grp * 16 + a, QB, g);
            cnt[(off / 4) % 32]++;
            cnt[(off / 4 + 1) % 32]++;
          }
          for (int b = 0; b < 32; b++) worst_w = cnt[b] > worst_w ? cnt[b] : worst_w;
        }
  for (int w = 0; w < 4; w++)
    for (int part = 0; part < 2; part++)
      for (int half = 0; half < 2; half++) {  // ds_read_b64_tr_b16: two 32-lane groups, bank = (addr / 4) % 64
        int cnt[64] = {0};
        for (int a = 0; a < 32; a++) {
          const int off = tr_addr(half * 32 + a, (w >> 1) * 32, part);
          cnt[(off / 4) % 64]++;
          cnt[(off / 4 + 1) % 64]++;
        }
        for (int b = 0; b < 64; b++) worst_r = cnt[b] > worst_r ? cnt[b] : worst_r;
      }
  printf("dS^T image: writes %d-way, transposed reads %d-way\n", worst_w, worst_r);
  CHECK(worst_w <= 2 && worst_r <= 1, "bank conflicts: writes %d-way, reads %d-way", worst_w, worst_r);

  // ---- 5. scratch order and accumulator coordinates
  {
    std::set<long> idx;
    std::set<std::pair<int, int>> qd;
    for (int w = 0; w < 4; w++)
      for (int l = 0; l < 64; l++)
        for (int r = 0; r < 16; r++) {
          idx.insert(acc_index(3, 17, 5, w, r >> 2, l, r & 3) - acc_index(3, 17, 5, 0, 0, 0, 0));
          int q, d;
          acc_coord(w, l, r, q, d);
          qd.insert({q, d});
        }
    CHECK(idx.size() == 4096 && *idx.begin() == 0 && *idx.rbegin() == 4095, "scratch order is not a bijection onto the tile");
    CHECK(qd.size() == 4096, "accumulator coordinates do not cover the tile once");
    CHECK(acc_index(3, 17, 6, 0, 0, 0, 0) - acc_index(3, 17, 5, 0, 0, 0, 0) == 4096, "tile stride");
  }
  if (fails) { printf("%d FAILED\n", fails); return 1; }
  printf("ok\n");
  return 0;
}
