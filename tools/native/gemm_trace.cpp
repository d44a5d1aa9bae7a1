// Per-workgroup timeline of the transposed-accumulator GEMM tiles (gemm3: 256 x 256, gemm4: 128 x 256) on the model's K = dim
// shapes.  Needs the diagnostic library (tools/build_trace_lib.sh).  Answers: where does the K-independent part of a launch go
// ("intercept 23-28 us") -- launch ramp, prologue, epilogue, the second round's start, the drain?
// Build + run: tools/native/run_gemm_trace.sh
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vbx.h"

extern "C" int vbx_debug_gemm2_trace(void*);
extern "C" int vbx_debug_gemm3_trace(void*);
extern "C" int vbx_debug_gemm4_trace(void*);

#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t r; memcpy(&r, &h, 2); return r; }
template <class T> static T* dev(const std::vector<T>& v) {
  T* p; HIPCHK(hipMalloc(&p, v.size() * sizeof(T) + 256));
  HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}
template <class T> static T* devfill(size_t n, int byte) { T* p; HIPCHK(hipMalloc(&p, n * sizeof(T) + 256)); HIPCHK(hipMemset(p, byte, n * sizeof(T))); return p; }
static std::vector<uint16_t> randn16(size_t n, float std, bool f16) {
  std::vector<uint16_t> v(n);
  for (size_t i = 0; i < n; i++) {
    float a = 0; for (int k = 0; k < 4; k++) a += (rand() / (float)RAND_MAX - 0.5f);
    a *= std * 1.732f;
    v[i] = f16 ? f2h(a) : f2bf(a);
  }
  return v;
}

static void stats(const char* what, std::vector<double> v) {
  if (v.empty()) return;
  std::sort(v.begin(), v.end());
  double s = 0; for (double x : v) s += x;
  printf("      %-22s n=%4zu  mean %6.2f  p0 %6.2f  p50 %6.2f  p90 %6.2f  p100 %6.2f\n", what, v.size(), s / v.size(), v.front(), v[v.size() / 2],
         v[v.size() * 9 / 10], v.back());
}

static void trace_one(const char* name, const vbx_gemm_desc& d, int path, int wgs) {
  vbx_gemm_select(path);
  unsigned long long* buf = devfill<unsigned long long>((size_t)8192 * 5, 0);
  for (int i = 0; i < 3; i++) if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm: %s\n", vbx_last_error()); exit(2); }
  HIPCHK(hipDeviceSynchronize());
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < 20; i++) vbx_gemm(&d, nullptr);
  HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  auto setter = path == 1 ? vbx_debug_gemm2_trace : (path == 2 ? vbx_debug_gemm3_trace : vbx_debug_gemm4_trace);
  setter(buf);
  vbx_gemm(&d, nullptr);
  HIPCHK(hipDeviceSynchronize());
  setter(nullptr);
  std::vector<unsigned long long> h((size_t)8192 * 5);
  HIPCHK(hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost));
  hipFree(buf);
  unsigned long long t0 = ~0ull;
  int n = 0;
  for (int i = 0; i < 8192; i++) if (h[i * 5]) { t0 = std::min(t0, h[i * 5]); n++; }
  printf("== %s, path %d (%s): %.1f us per launch back to back; %d workgroups traced (expected %d)\n", name, path, path == 1 ? "128x128, 3 per CU" : (path == 2 ? "gemm3 256x256" : "gemm4 128x256"),
         ms * 50.f, n, wgs);
  // classify by start time: first wave of workgroups (started within 3 us of the first) vs later ones
  std::vector<double> st[2], pro[2], loop[2], epi[2], life[2], endt[2];
  double span = 0;
  std::vector<std::pair<double, double>> iv;
  for (int i = 0; i < 8192; i++) {
    if (!h[i * 5]) continue;
    const double s = (h[i * 5] - t0) / 100.0, p = h[i * 5 + 1] ? (h[i * 5 + 1] - t0) / 100.0 : s, l = (h[i * 5 + 2] - t0) / 100.0, e = (h[i * 5 + 3] - t0) / 100.0;
    const int r = s > 3.0;
    st[r].push_back(s); pro[r].push_back(p - s); loop[r].push_back(l - p); epi[r].push_back(e - l); life[r].push_back(e - s); endt[r].push_back(e);
    span = std::max(span, e);
    iv.push_back({s, e});
  }
  printf("   span first start -> last end: %.2f us\n", span);
  {  // which launch slots (blockIdx / 256) share a CU in the first wave?
    int hist[4][4] = {};
    std::vector<std::vector<int>> on(8 * 8 * 16);
    for (int i = 0; i < 8192; i++) {
      if (!h[i * 5] || (h[i * 5] - t0) / 100.0 > 3.0) continue;
      const unsigned long long r = h[i * 5 + 4];
      const int cu = (r >> 8) & 0xF, se = (r >> 13) & 0x7, xcc = (r >> 32) & 0xF;
      on[(xcc * 8 + se) * 16 + cu].push_back(i);
    }
    int cus = 0;
    for (auto& v : on) {
      if (v.empty()) continue;
      cus++;
      for (size_t a = 0; a < v.size(); a++) for (size_t b = a + 1; b < v.size(); b++) hist[std::min(3, v[a] >> 8)][std::min(3, v[b] >> 8)]++;
    }
    printf("   first wave on %d CUs; co-resident pairs by launch slot (blockIdx/256): 0-0 %d, 0-1 %d, 0-2 %d, 1-1 %d, 1-2 %d, 2-2 %d\n", cus, hist[0][0],
           hist[0][1] + hist[1][0], hist[0][2] + hist[2][0], hist[1][1], hist[1][2] + hist[2][1], hist[2][2]);
  }
  for (int r = 0; r < 2; r++) {
    if (st[r].empty()) continue;
    printf("   %s\n", r ? "later workgroups" : "first wave of workgroups");
    stats("start", st[r]); stats("prologue (entry->k0)", pro[r]); stats("k-loop", loop[r]); stats("epilogue", epi[r]); stats("lifetime", life[r]); stats("end", endt[r]);
  }
  // residency histogram
  printf("   t(us)    :"); for (double x = 0; x < span; x += 2.5) printf(" %4.0f", x); printf("\n   resident :");
  for (double x = 0; x < span; x += 2.5) { int c = 0; for (auto& p : iv) if (p.first <= x && p.second > x) c++; printf(" %4d", c); }
  printf("\n");
}

int main(int argc, char** argv) {
  srand(11);
  const int Bt = 8, Np = 1040, M = Bt * Np, D = 512, H = 16, I = 1024, Fp = 1408;
  auto A512h = dev(randn16((size_t)M * D, 1.0f, true));
  auto A512b = dev(randn16((size_t)M * D, 1.0f, false));
  auto Wqkvh = dev(randn16((size_t)3 * I * D, 0.044f, true));
  auto Wqkvb = dev(randn16((size_t)3 * I * D, 0.044f, false));
  auto W1h = dev(randn16((size_t)2 * Fp * D, 0.044f, true));
  std::vector<float> fb(4096, 0.01f), tab((size_t)Np * 32, 0.7f), gam(I, 1.0f);
  float *bias = dev(fb), *rc = dev(tab), *rs = dev(tab), *qg = dev(gam), *kg = dev(gam);
  const size_t hs = (size_t)Bt * H * Np * 64;
  uint16_t *q16 = devfill<uint16_t>(hs, 0), *k16 = devfill<uint16_t>(hs, 0), *qb = devfill<uint16_t>(hs, 0), *kb = devfill<uint16_t>(hs, 0), *v = devfill<uint16_t>(hs, 0),
           *v16 = devfill<uint16_t>(hs, 0);
  float *qrn = devfill<float>((size_t)Bt * H * Np, 0), *krn = devfill<float>((size_t)Bt * H * Np, 0);
  uint16_t *G = devfill<uint16_t>((size_t)M * Fp, 0), *Gb = devfill<uint16_t>((size_t)M * Fp, 0), *H1 = devfill<uint16_t>((size_t)M * 2 * Fp, 0);
  uint16_t* Cb = devfill<uint16_t>((size_t)M * 3 * I, 0);
  auto base = [&](int mode, int epi, int N, int K, const void* A, int lda, const void* B, int ldb) {
    vbx_gemm_desc d{}; d.mode = mode; d.epilogue = epi; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.A = A; d.B = B; return d;
  };
  vbx_gemm_desc plain = base(VBX_GEMM_NT, VBX_EPI_BF16, 3 * I, D, A512b, D, Wqkvb, D); plain.C = Cb; plain.ldc = 3 * I;
  vbx_gemm_desc qkv = base(VBX_GEMM_NT, VBX_EPI_QKV, 3 * I, D, A512h, D, Wqkvh, D);
  qkv.f16 = 1; qkv.Np = Np; qkv.H = H; qkv.qk_scale = 8.f; qkv.q_gamma = qg; qkv.k_gamma = kg; qkv.rot_cos = rc; qkv.rot_sin = rs;
  qkv.q16 = q16; qkv.k16 = k16; qkv.v16 = v16;
  vbx_gemm_desc qkvt = qkv; qkvt.qb = qb; qkvt.kb = kb; qkvt.v = v; qkvt.q_rnorm = qrn; qkvt.k_rnorm = krn;
  vbx_gemm_desc ff = base(VBX_GEMM_NT, VBX_EPI_GEGLU, 2 * Fp, D, A512h, D, W1h, D);
  ff.f16 = 1; ff.C = G; ff.ldc = Fp; ff.bias = bias;
  vbx_gemm_desc fft = ff; fft.C2 = H1; fft.C3 = Gb;
  if (argc > 1 && !strcmp(argv[1], "alias")) {
    // Is the k-loop of the one-round N = dim tile waiting for L2 MISSES of its activation panel?  Same GEMM with the rows of A
    // aliased onto a few KB (lda = 8: every A piece is an L1 / L2 hit) against the real layout.
    const int K = 3072;
    auto Abig = dev(randn16((size_t)M * K, 1.0f, false));
    auto Wb = dev(randn16((size_t)K * D, 0.03f, false));
    uint16_t* Co = devfill<uint16_t>((size_t)M * D, 0);
    for (int lda : {K, 8, K, 8}) {
      vbx_gemm_desc d = base(VBX_GEMM_NN, VBX_EPI_BF16, D, K, Abig, lda, Wb, D); d.C = Co; d.ldc = D;
      vbx_gemm_select(1);
      for (int i = 0; i < 3; i++) if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm: %s\n", vbx_last_error()); return 2; }
      hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < 20; i++) vbx_gemm(&d, nullptr);
      HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
      float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      printf("dgrad to_qkv shape (NN, M=8320 N=512 K=3072), lda = %4d: %6.1f us\n", lda, ms * 50.f);
    }
    // the same for the 256 x 256 tile at a K-loop-dominated shape (N = 3072, K = 4096: A = 68 MB)
    {
      const int N2 = 3072, K2 = 4096;
      auto X = dev(randn16((size_t)M * K2, 1.0f, false)); auto Y = dev(randn16((size_t)N2 * K2, 0.03f, false)); uint16_t* Z = devfill<uint16_t>((size_t)M * N2, 0);
      for (int path = 1; path <= 2; path++)
        for (int lda : {K2, 8}) {
          vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_BF16, N2, K2, X, lda, Y, K2); d.C = Z; d.ldc = N2;
          vbx_gemm_select(path);
          for (int i = 0; i < 3; i++) vbx_gemm(&d, nullptr);
          hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
          HIPCHK(hipEventRecord(e0, nullptr));
          for (int i = 0; i < 10; i++) vbx_gemm(&d, nullptr);
          HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
          float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
          printf("NT M=8320 N=3072 K=4096 path %d, lda = %4d: %6.1f us\n", path, lda, ms * 100.f);
        }
      // both operands aliased
      for (int path = 1; path <= 2; path++) {
        vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_BF16, N2, K2, X, 8, Y, 8); d.C = Z; d.ldc = N2;
        vbx_gemm_select(path);
        for (int i = 0; i < 3; i++) vbx_gemm(&d, nullptr);
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 10; i++) vbx_gemm(&d, nullptr);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        printf("NT M=8320 N=3072 K=4096 path %d, lda = ldb = 8: %6.1f us\n", path, ms * 100.f);
      }
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "time")) {  // back-to-back launch times of all three tiles (VBX_GEMM_STAGGER A/B: one process per value)
    struct { const char* n; vbx_gemm_desc* d; } L[5] = {{"plain bf16 N=3072", &plain}, {"to_qkv eval", &qkv}, {"to_qkv train", &qkvt}, {"ff_in eval", &ff}, {"ff_in train", &fft}};
    printf("stagger %s us:", getenv("VBX_GEMM_STAGGER") ? getenv("VBX_GEMM_STAGGER") : "0");
    for (auto& l : L) {
      printf("  %s", l.n);
      for (int path = 1; path <= 3; path++) {
        vbx_gemm_select(path);
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
          for (int i = 0; i < 3; i++) vbx_gemm(l.d, nullptr);
          hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
          HIPCHK(hipEventRecord(e0, nullptr));
          for (int i = 0; i < 20; i++) vbx_gemm(l.d, nullptr);
          HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
          float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
          best = fminf(best, ms * 50.f);
        }
        printf(" %5.1f", best);
      }
      printf(" |");
    }
    printf("   (128^2 / gemm3 / gemm4)\n");
    return 0;
  }
  for (int path = 1; path <= 3; path++) {
    if (argc > 1 && atoi(argv[1]) && atoi(argv[1]) != path) continue;
    const int t = path == 2 ? 33 : (path == 1 ? 130 : 65);
    trace_one("N=3072 K=512 plain bf16 epilogue", plain, path, t * 12);
    trace_one("to_qkv eval", qkv, path, t * 12);
    trace_one("to_qkv train", qkvt, path, t * 12);
    trace_one("ff_in eval", ff, path, t * 11);
    trace_one("ff_in train", fft, path, t * 11);
  }
  return 0;
}
