// Per-workgroup time line of the weight-stationary GEMM (diagnostic library built with -DVBX_G5_TRACE: tools/native/g5_trace.sh).
// Stamps of wave 0 (s_memtime, 100 MHz): 0 entry, 1 weights loaded, then per block j: 2+4j before the vmcnt wait, 3+4j after it,
// 4+4j after the barrier, 5+4j after the DMA issue (the MFMA + epilogue phase runs until the next 2+4(j+1)); last: epilogue-only phase.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../include/vbx.h"
extern "C" int vbx_debug_gemm5_trace(void* buf);
#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t r; memcpy(&r, &h, 2); return r; }
int main(int argc, char** argv) {
  const int train = argc > 1 ? atoi(argv[1]) : 0, geglu = argc > 2 ? atoi(argv[2]) : 0;
  const int Bt = 8, Np = 1040, M = Bt * Np, D = 512, H = 16, I = 1024, Fp = 1408;
  std::vector<uint16_t> a((size_t)M * D), w((size_t)3 * I * D);
  for (auto& v : a) v = f2h((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : w) v = f2h((rand() % 2001 - 1000) / 20000.f);
  uint16_t *dA, *dW; HIPCHK(hipMalloc(&dA, a.size() * 2)); HIPCHK(hipMalloc(&dW, w.size() * 2));
  HIPCHK(hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dW, w.data(), w.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> tab((size_t)Np * 32, 0.7f), gam(I, 1.0f), fb(4096, 0.01f);
  float *rc, *rs, *qg, *bias; HIPCHK(hipMalloc(&rc, tab.size() * 4)); HIPCHK(hipMalloc(&rs, tab.size() * 4)); HIPCHK(hipMalloc(&qg, gam.size() * 4)); HIPCHK(hipMalloc(&bias, fb.size() * 4));
  HIPCHK(hipMemcpy(rc, tab.data(), tab.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(rs, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(qg, gam.data(), gam.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(bias, fb.data(), fb.size() * 4, hipMemcpyHostToDevice));
  const size_t hs = (size_t)Bt * H * Np * 64;
  uint16_t* outs[6]; for (auto& o : outs) HIPCHK(hipMalloc(&o, hs * 2 * 2));
  float* rn[2]; for (auto& o : rn) HIPCHK(hipMalloc(&o, (size_t)Bt * H * Np * 4));
  vbx_gemm_desc d{};
  d.mode = VBX_GEMM_NT; d.M = M; d.K = D; d.lda = D; d.ldb = D; d.A = dA; d.B = dW; d.f16 = 1;
  if (!geglu) {
    d.epilogue = VBX_EPI_QKV; d.N = 3 * I; d.Np = Np; d.H = H; d.qk_scale = 8.f; d.q_gamma = qg; d.k_gamma = qg; d.rot_cos = rc; d.rot_sin = rs; d.q_prescale = 14.4f;
    d.q16 = outs[0]; d.k16 = outs[1]; d.v16 = outs[2];
    if (train) { d.qb = outs[3]; d.kb = outs[4]; d.v = outs[5]; d.q_rnorm = rn[0]; d.k_rnorm = rn[1]; }
  } else {
    d.epilogue = VBX_EPI_GEGLU; d.N = 2 * Fp; d.C = outs[0]; d.ldc = Fp; d.bias = bias;
    if (train) { d.C2 = outs[1]; d.C3 = outs[2]; }
  }
  vbx_gemm_select(4);
  const int nwg = 256;
  unsigned long long* tb; HIPCHK(hipMalloc(&tb, (size_t)nwg * 64 * 8)); HIPCHK(hipMemset(tb, 0, (size_t)nwg * 64 * 8));
  for (int i = 0; i < 3; i++) vbx_gemm(&d, nullptr);
  HIPCHK(hipDeviceSynchronize());
  vbx_debug_gemm5_trace(tb);
  if (vbx_gemm(&d, nullptr)) { printf("%s\n", vbx_last_error()); return 2; }
  HIPCHK(hipDeviceSynchronize());
  vbx_debug_gemm5_trace(nullptr);
  std::vector<unsigned long long> t((size_t)nwg * 64);
  HIPCHK(hipMemcpy(t.data(), tb, t.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; for (int g = 0; g < nwg; g++) if (t[g * 64]) t0 = std::min(t0, t[g * 64]);
  printf("stamps in us from the first workgroup's entry (100 MHz s_memtime); wg: entry, W loaded, then per block [wait, barrier, issue, phase]\n");
  for (int g : {0, 1, 21, 100, 251}) {
    const unsigned long long* r = &t[(size_t)g * 64];
    printf("wg %3d: entry %.2f  W %.2f |", g, (r[0] - t0) / 100.0, (r[1] - r[0]) / 100.0);
    for (int j = 0; j < 14 && r[2 + 4 * j + 1]; j++) {
      const double wait = (r[3 + 4 * j] - r[2 + 4 * j]) / 100.0, bar = (r[4 + 4 * j] - r[3 + 4 * j]) / 100.0, iss = (r[5 + 4 * j] - r[4 + 4 * j]) / 100.0;
      const double ph = r[2 + 4 * (j + 1)] ? (r[2 + 4 * (j + 1)] - r[5 + 4 * j]) / 100.0 : 0;
      printf(" [%.2f %.2f %.2f %.2f]", wait, bar, iss, ph);
    }
    int last = 0; for (int i = 0; i < 64; i++) if (r[i]) last = i;
    printf("  end %.2f\n", (r[last] - t0) / 100.0);
  }
  // averages over all workgroups
  double sw = 0, sb = 0, si = 0, sp = 0, sW = 0; long n = 0; int ng = 0;
  for (int g = 0; g < nwg; g++) { const unsigned long long* r = &t[(size_t)g * 64]; if (!r[0]) continue; ng++; sW += (r[1] - r[0]) / 100.0;
    for (int j = 0; j < 14 && r[2 + 4 * j + 1] && r[2 + 4 * (j + 1)]; j++) { sw += (r[3 + 4 * j] - r[2 + 4 * j]) / 100.0; sb += (r[4 + 4 * j] - r[3 + 4 * j]) / 100.0; si += (r[5 + 4 * j] - r[4 + 4 * j]) / 100.0; sp += (r[2 + 4 * (j + 1)] - r[5 + 4 * j]) / 100.0; n++; } }
  printf("mean over %d workgroups: W load %.2f us; per block: vmcnt wait %.3f, barrier %.3f, DMA issue %.3f, MFMA + epilogue phase %.3f us\n", ng, sW / ng, sw / n, sb / n, si / n, sp / n);
  return 0;
}
