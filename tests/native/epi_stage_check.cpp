// Host-side check of the row-staged epilogue layout (csrc/epi_stage_layout.hpp), for NJ = 8 (a wave's 128 columns) and NJ = 4
// (the 64 GEGLU output columns):
//   1. writing every lane's 8 bytes of every (il, j) block and reading the image back with the row-major 16-byte reads yields, for
//      each (pass row, chunk), exactly the 8 columns chunk*8 .. +7 of that row, in order, every (row, chunk) exactly once;
//   2. bank census: every ds_write_b64 instruction is at most 2-way (4 groups of 16 consecutive lanes, 32 banks of 4 bytes), every
//      ds_read_b128 instruction conflict free (lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32; 64 banks of 4 bytes);
//   3. the image of a pass fits the per-wave staging budget (two passes of NJ = 8, or 4 + 4 + 8 KiB for the GEGLU epilogue).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>
#include "../../voicebox-pytorch_amd/csrc/epi_stage_layout.hpp"

static int bad = 0;
#define CHECK(c, ...) do { if (!(c)) { if (bad < 20) { printf(__VA_ARGS__); printf("\n"); } bad++; } } while (0)

static int worst_way(const std::vector<int>& addrs, const std::vector<std::vector<int>>& groups, int nbanks, int dwords) {
  int worst = 0;
  for (auto& grp : groups) {
    std::map<int, std::set<int>> banks;
    for (int l : grp)
      for (int d = 0; d < dwords; d++) banks[(addrs[l] / 4 + d) % nbanks].insert(addrs[l] / 4 + d);
    for (auto& kv : banks) worst = std::max(worst, (int)kv.second.size());
  }
  return worst;
}

int main() {
  std::vector<std::vector<int>> wgroups, rgroups;
  for (int g = 0; g < 4; g++) { std::vector<int> v; for (int l = 0; l < 16; l++) v.push_back(g * 16 + l); wgroups.push_back(v); }
  for (int half = 0; half < 2; half++) {
    std::vector<int> a, b;
    for (int l : {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}) a.push_back(l + 32 * half);
    for (int l : {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}) b.push_back(l + 32 * half);
    rgroups.push_back(a); rgroups.push_back(b);
  }
  for (int NJ : {8, 4}) {
    const int bytes = 32 * epst::stride(NJ);
    std::vector<uint16_t> img(bytes / 2, 0xffff);
    std::vector<int> writes(bytes / 8, 0);
    // element id = row * 256 + column (row 0..31, column 0..NJ*16-1)
    for (int il = 0; il < 2; il++)
      for (int j = 0; j < NJ; j++) {
        std::vector<int> addrs(64);
        for (int lane = 0; lane < 64; lane++) {
          const int m = lane & 15, g = lane >> 4;
          const int off = epst::put_byte(NJ, il, j, lane);
          CHECK(off >= 0 && off + 8 <= bytes && off % 8 == 0, "NJ %d put out of range / misaligned: il %d j %d lane %d -> %d", NJ, il, j, lane, off);
          addrs[lane] = off;
          writes[off / 8]++;
          for (int r = 0; r < 4; r++) img[off / 2 + r] = (uint16_t)((il * 16 + m) * 256 + j * 16 + 4 * g + r);
        }
        const int w = worst_way(addrs, wgroups, 32, 2);
        CHECK(w <= 2, "NJ %d ds_write_b64 (il %d, j %d) is %d-way", NJ, il, j, w);
      }
    for (int w : writes) CHECK(w == 1, "NJ %d: an 8-byte slot written %d times", NJ, w);
    std::set<int> seen;
    for (int it = 0; it < epst::its(NJ); it++) {
      std::vector<int> addrs(64);
      for (int lane = 0; lane < 64; lane++) {
        const int row = epst::get_row(NJ, it, lane), ch = epst::get_chunk(NJ, lane), off = epst::get_byte(NJ, it, lane);
        CHECK(off >= 0 && off + 16 <= bytes && off % 16 == 0, "NJ %d get out of range / misaligned", NJ);
        CHECK(row >= 0 && row < 32 && ch >= 0 && ch < epst::cpr(NJ), "NJ %d bad (row, chunk)", NJ);
        addrs[lane] = off;
        CHECK(seen.insert(row * 64 + ch).second, "NJ %d (row %d, chunk %d) read twice", NJ, row, ch);
        for (int e = 0; e < 8; e++)
          CHECK(img[off / 2 + e] == (uint16_t)(row * 256 + ch * 8 + e), "NJ %d it %d lane %d element %d: got %u want row %d col %d", NJ, it, lane, e,
                img[off / 2 + e], row, ch * 8 + e);
      }
      const int w = worst_way(addrs, rgroups, 64, 4);
      CHECK(w == 1, "NJ %d ds_read_b128 (it %d) is %d-way", NJ, it, w);
    }
    CHECK((int)seen.size() == 32 * epst::cpr(NJ), "NJ %d: %zu of %d (row, chunk) pairs read", NJ, seen.size(), 32 * epst::cpr(NJ));
  }
  CHECK(2 * 32 * epst::stride(8) <= 16384, "two NJ = 8 passes exceed the 16 KiB per-wave stage");
  CHECK(2 * 32 * epst::stride(4) + 32 * epst::stride(8) <= 16384, "GEGLU stage (G | Gb | H1) exceeds 16 KiB");
  printf(bad ? "epi stage layout check FAILED: %d problems\n" : "epi stage layout check ok\n", bad);
  return bad != 0;
}
