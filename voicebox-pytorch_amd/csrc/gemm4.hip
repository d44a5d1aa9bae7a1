// gemm4: 128 x 256 x 32 MFMA GEMM tile for gfx950, 4 waves (2 x 2, each 64 x 128), 3-stage LDS-DMA ring (72 KiB), TWO
// workgroups per CU, transposed accumulators + register epilogues (gemm_epi3.hpp).
//
// For the model's wide, SHORT-K GEMMs (to_qkv, FeedForward-in and the dgrads into their widths: K = dim = 512, N = 1024-3072).
// These write 50-100 MB per launch for ~25 GFLOP -- close to the machine balance -- so the output stores of one tile must run
// under the MFMAs of another.  The 256 x 256 tile of gemm3.hip has one workgroup per CU: its k-loop (8 k-tiles at K = 512) and
// its epilogue alternate, with the matrix pipe idle during every epilogue and the memory system idle during every k-loop
// (measured: to_qkv 54-72 us against 52-56 us for the 128 x 128 kernel; at 8192^3 the same kernel reaches 1293 TFLOP/s).
// Here two independent workgroups share a CU (one wave of each per SIMD): while one stores, the other multiplies; each wave
// still owns 64 x 128 outputs (12 ds_read_b128 per 32 MFMAs, as in gemm3) and 128-row tiles divide M = 8 x 1040 exactly.
// k-loop = gemm.hip's v2 protocol (validated on hardware): counted vmcnt -> raw s_barrier -> LDS-DMA of stage t+2 ->
// fragment reads (inline asm) -> lgkmcnt(0) -> MFMAs.
#include "common.hpp"
#include "gemm4_layout.hpp"
#include "gemm_epi3.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

using namespace g4;
using namespace gepi;

struct G4Params {
  const u16* A;
  const u16* B;
  int M, N, K;
  long lda, ldb;
  int stagger;  // start-phase stagger of the second workgroup per CU in 10 ns ticks (common.hpp::stagger_wait); 0 = off
  int late_dma; // a k-tile's DMAs issued behind the fragment reads (default; VBX_GEMM_LATE_DMA=0: in front of them, A/B)
};

__device__ uint4 g4_zero_page[4];
}  // namespace
#ifdef VBX_GEMM_TRACE
extern "C" int vbx_debug_gemm4_trace(void* buf) {  // diagnostic build only: buf = [workgroups][5] u64, null to stop
  return hipMemcpyToSymbol(HIP_SYMBOL(gepi::g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
namespace {

// LDS-DMA sources of one operand.  K-contiguous: piece q covers rows +64q of the stage (A: q < 2, B: q < 4).
// K-strided: per 128-wide image two pieces (k rows +16); B has two images (columns +128).
template <int MODE, bool IS_A>
struct Stage4 {
  static constexpr int ND = IS_A ? A_DMAS : B_DMAS;
  const u16* base;
  long ld;
  int outer0, kq, olim;
  VBX_DEV void init(const u16* __restrict__ X, long ld_, int o0, int olim_, int tid) {
    int o, k;
    if (MODE == 0) kc_slot(tid, o, k); else ks_slot(tid, o, k);
    outer0 = o0 + o; kq = k; olim = olim_; ld = ld_;
    base = (MODE == 0) ? X + (long)outer0 * ld + k : X + (long)k * ld + outer0;
  }
  template <bool FULL>
  VBX_DEV void issue(char* dst, int k0, int kend, int wave) const {
#pragma unroll
    for (int q = 0; q < ND; q++) {
      int dout, dk;  // outer / k displacement of piece q
      if (MODE == 0) { dout = 64 * q; dk = 0; }
      else { dout = 128 * (q >> 1); dk = 16 * (q & 1); }
      const u16* src = (MODE == 0) ? base + (long)dout * ld + k0 : base + (long)(k0 + dk) * ld + dout;
      if (!FULL) {
        const bool ok = (outer0 + dout < olim) && (k0 + kq + dk < kend);
        if (!ok) src = reinterpret_cast<const u16*>(g4_zero_page);
      }
      char* wave_dst = dst + (q * THREADS + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)wave_dst, 16, 0, 0);
    }
  }
};

VBX_DEV unsigned lds_u32(const char* p) { return (unsigned)(size_t)LDS_PTR(char, p); }

#define G4_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define G4_DS_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")

struct RawFrag4 {
  bf16x8 v;
  s16x4 lo, hi;
};
template <int MODE>
VBX_DEV bf16x8 frag_value4(const RawFrag4& f) {
  if (MODE == 0) return f.v;
  s16x8 r = {f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
  return __builtin_bit_cast(bf16x8, r);
}
// per-lane fragment addresses of one operand inside stage 0 (stage and sub-tile offsets are DS immediates)
template <int MODE, int NF>
struct Frag4 {
  unsigned a[MODE == 0 ? 1 : NF];
  VBX_DEV void init(const char* op_base, int o_w, int lane) {  // o_w: first outer index of the wave inside the operand stage
    if (MODE == 0) {
      a[0] = lds_u32(op_base + kc_frag_byte(o_w, lane));
    } else {
#pragma unroll
      for (int f = 0; f < NF; f++) {
        const int o = o_w + f * 16;
        a[f] = lds_u32(op_base + (o >> 7) * KS_IMAGE + ks_frag_byte(o & 127, lane, 0));
      }
    }
  }
  template <int OFF, int F>
  VBX_DEV void read(RawFrag4& out) const {
    if constexpr (MODE == 0) {
      G4_DS_B128(out.v, a[0], OFF + F * 1024);  // 16 rows further = 8 line pairs = 1024 bytes, same swizzle phase
    } else {
      G4_DS_TR(out.lo, a[F], OFF);
      G4_DS_TR(out.hi, a[F], OFF + 1024);
    }
  }
};

template <int MA, int MB, class Epi, bool F16>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(G4Params p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GEMM_TRACE_DECL();
  if (p.stagger && blockIdx.x >= 256 && blockIdx.x < 512) stagger_wait(1, p.stagger);  // 2 workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: every XCD gets one contiguous chunk of the tile sequence (n fastest)
  const int T = gridDim.x, xcd = blockIdx.x & 7, qi = blockIdx.x >> 3;
  const int q8 = T >> 3, r8 = T & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + qi;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kend = p.K;
  const int nt = (kend + BK - 1) / BK;
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);

  Stage4<MA, true> sa;
  Stage4<MB, false> sb;
  sa.init(p.A, p.lda, m0, p.M, tid);
  sb.init(p.B, p.ldb, n0, p.N, tid);
  Frag4<MA, 4> fa;
  Frag4<MB, 8> fb;
  fa.init(smem, wr * 64, lane);
  fb.init(smem + A_BYTES, wc * 128, lane);  // B stage base inside slot 0: the reads below add only the slot offset SO

  Acc acc;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int t) {  // k-tile t -> ring slot t % 3
    char* dst = smem + (t % NST) * STAGE;
    const int k0 = t * BK;
    if (interior && k0 + BK <= kend) {
      sa.template issue<true>(dst, k0, kend, wave);
      sb.template issue<true>(dst + A_BYTES, k0, kend, wave);
    } else {
      sa.template issue<false>(dst, k0, kend, wave);
      sb.template issue<false>(dst + A_BYTES, k0, kend, wave);
    }
  };
  if (nt > 0) stage(0);
  if (nt > 1) stage(1);

  auto step = [&](auto stg_c, int t) {
    constexpr int STG = decltype(stg_c)::value;
    constexpr int SO = STG * STAGE;
    // this thread's 6 DMAs of k-tile t have landed once at most one younger stage is pending
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // k-tile t visible to all waves; everyone is done reading k-tile t-1
    if (!p.late_dma && t + 2 < nt) stage(t + 2);  // into the slot k-tile t-1 used
    RawFrag4 af[4], bf[8];
    fa.template read<SO, 0>(af[0]); fa.template read<SO, 1>(af[1]); fa.template read<SO, 2>(af[2]); fa.template read<SO, 3>(af[3]);
    fb.template read<SO, 0>(bf[0]); fb.template read<SO, 1>(bf[1]); fb.template read<SO, 2>(bf[2]); fb.template read<SO, 3>(bf[3]);
    fb.template read<SO, 4>(bf[4]); fb.template read<SO, 5>(bf[5]); fb.template read<SO, 6>(bf[6]); fb.template read<SO, 7>(bf[7]);
    if (p.late_dma && t + 2 < nt) stage(t + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = mfma16<F16>(frag_value4<MB>(bf[j]), frag_value4<MA>(af[i]), acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nt; t += 3) {
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1 < nt) step(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < nt) step(std::integral_constant<int, 2>{}, t + 2);
  }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();  // every wave is done with the ring: it becomes the epilogues' row-staging space
  __builtin_amdgcn_sched_barrier(0);
  GEMM_TRACE_MARK(gtr2);
  epi(acc, m0 + wr * 64, n0 + wc * 128, lane, 0, p.M, p.N, 32, lds_u32(smem + wave * EPI_STAGE_BYTES));
  GEMM_TRACE_END();
}

template <int MA, int MB, bool F16 = false, class Epi>
int launch4(const G4Params& p, const Epi& epi, hipStream_t st) {
  auto kern = gemm4_kernel<MA, MB, Epi, F16>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vbx_set_error("gemm4: cannot opt in to %d bytes of LDS: %s", LDS_BYTES, hipGetErrorString(e));
      return (int)e;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(p.M, BM) * cdiv(p.N, BN)), dim3(THREADS), LDS_BYTES, st, p, epi);
  VBX_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Same contract as vbx_gemm (include/vbx.h) for NT / NN descriptors; VBX_EUNSUPPORTED = not served here.
int vbx_gemm4(const vbx_gemm_desc* d, hipStream_t st) {
  if (!d || !d->A || !d->B || d->M <= 0 || d->N <= 0 || d->K <= 0) return VBX_EUNSUPPORTED;
  if (d->lda % 8 || d->ldb % 8 || d->N % 8 || d->K % 8) return VBX_EUNSUPPORTED;
  if (d->mode != VBX_GEMM_NT && d->mode != VBX_GEMM_NN) return VBX_EUNSUPPORTED;
  static const int stagger = getenv("VBX_GEMM_STAGGER") ? (int)(atof(getenv("VBX_GEMM_STAGGER")) * 100.0) : 0;
  static const int late = getenv("VBX_GEMM_LATE_DMA") ? atoi(getenv("VBX_GEMM_LATE_DMA")) : 1;
  G4Params p{(const u16*)d->A, (const u16*)d->B, d->M, d->N, d->K, d->lda, d->ldb, stagger, late};
  switch (d->epilogue) {
    case VBX_EPI_BF16: {
      if (!d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
      if (d->delta) {  // dgrad of to_out with the attention backward's delta as a by-product (gemm_epi3.hpp::Epi3BF16Delta)
        if (d->mode != VBX_GEMM_NN || d->bias || !d->delta_o || d->H <= 0 || d->Np <= 0 || d->N != d->H * 64 || d->M % d->Np)
          return VBX_EUNSUPPORTED;
        Epi3BF16Delta e{(u16*)d->C, d->ldc, (const u16*)d->delta_o, d->delta, d->H, d->Np};
        return launch4<0, 1>(p, e, st);
      }
      Epi3BF16 e{(u16*)d->C, d->ldc, d->bias};
      if (d->mode == VBX_GEMM_NT && !d->f16) return launch4<0, 0>(p, e, st);
      if (d->mode == VBX_GEMM_NN) return launch4<0, 1>(p, e, st);
      break;
    }
    case VBX_EPI_F32: {
      if (!d->C || d->ldc % 8) return VBX_EUNSUPPORTED;
      Epi3F32 e{(float*)d->C, d->ldc, d->bias, d->resid, (u16*)d->C2};
      if (d->mode == VBX_GEMM_NT && d->f16) return launch4<0, 0, true>(p, e, st);
      if (d->mode == VBX_GEMM_NT) return launch4<0, 0>(p, e, st);
      if (d->mode == VBX_GEMM_NN) return launch4<0, 1>(p, e, st);
      break;
    }
    case VBX_EPI_QKV: {
      if (d->mode != VBX_GEMM_NT || d->H <= 0 || d->H % 2 || d->N != 3 * d->H * 64 || d->Np <= 0 || d->M % d->Np) return VBX_EUNSUPPORTED;
      if (!d->q16 || !d->k16 || !(d->v || d->v16) || !d->rot_cos || !d->rot_sin) return VBX_EUNSUPPORTED;
      if (d->qk_scale > 0.f && !(d->q_gamma && d->k_gamma)) return VBX_EUNSUPPORTED;
      Epi3QKV e{d->Np, d->H, d->qk_scale, d->q_gamma, d->k_gamma, d->rot_cos, d->rot_sin,
                (u16*)d->q16, (u16*)d->k16, (u16*)d->qb, (u16*)d->kb, (u16*)d->v, d->q_rnorm, d->k_rnorm, (u16*)d->v16,
                d->q_prescale > 0.f ? d->q_prescale : 1.0f};
      if (d->f16) return launch4<0, 0, true>(p, e, st);
      return launch4<0, 0>(p, e, st);
    }
    case VBX_EPI_GEGLU: {
      if (d->mode != VBX_GEMM_NT || d->N % 128 || !d->bias || !d->C) return VBX_EUNSUPPORTED;
      Epi3GEGLU e{(u16*)d->C, d->ldc, d->bias, (u16*)d->C2, d->N, (u16*)d->C3, d->f16};
      if (d->f16) return launch4<0, 0, true>(p, e, st);
      return launch4<0, 0>(p, e, st);
    }
    default: break;
  }
  return VBX_EUNSUPPORTED;
}
