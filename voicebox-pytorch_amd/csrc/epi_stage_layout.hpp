// LDS image of the row-staged GEMM epilogues (gemm_epi3.hpp::RowStage) as plain index functions, shared by the kernels and by the
// host-side check (tests/native/epi_stage_check.cpp).  A wave transposes a 32-row pass of its transposed accumulators -- lane
// (m = lane & 15, g = lane >> 4) owns columns 16 j + 4 g .. + 3 of row 16 il + m, four 16-bit values = 8 bytes -- into a row-major
// image and reads it back 16 bytes per lane, CPR = 2 NJ chunks per row, 64 / CPR rows per instruction.
#pragma once
#ifdef __HIPCC__
#define EPST_HD __host__ __device__ inline
#else
#define EPST_HD inline
#endif

namespace epst {

// NJ = 16-column blocks per row: 8 (a wave's 128 columns, 256-byte rows) or 4 (64 columns, 128-byte rows)
EPST_HD int stride(int NJ) { return NJ * 32; }   // bytes per staged row
EPST_HD int cpr(int NJ) { return 2 * NJ; }       // 16-byte chunks per row
EPST_HD int rpi(int NJ) { return 64 / cpr(NJ); } // rows per read-back instruction
EPST_HD int its(int NJ) { return 32 / rpi(NJ); } // read-back instructions per 32-row pass
// 16-byte chunk c of row r lives at chunk c ^ (r & (CPR - 1)): the ds_write_b64 of one (il, j) is 2-way, the row-major
// ds_read_b128 conflict free
EPST_HD int chunk_byte(int NJ, int r, int c) { return r * stride(NJ) + ((c ^ (r & (cpr(NJ) - 1))) << 4); }
// where lane's 8 bytes of block (il, j) go
EPST_HD int put_byte(int NJ, int il, int j, int lane) {
  const int m = lane & 15, g = lane >> 4;
  return chunk_byte(NJ, il * 16 + m, j * 2 + (g >> 1)) | ((g & 1) << 3);
}
// what read-back instruction `it` hands to a lane: row (inside the pass), chunk (8 columns), and its byte offset
EPST_HD int get_row(int NJ, int it, int lane) { return it * rpi(NJ) + lane / cpr(NJ); }
EPST_HD int get_chunk(int NJ, int lane) { return lane % cpr(NJ); }
EPST_HD int get_byte(int NJ, int it, int lane) { return chunk_byte(NJ, get_row(NJ, it, lane), get_chunk(NJ, lane)); }

}  // namespace epst
