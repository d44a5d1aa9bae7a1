// Stand-alone (no Python / torch) GPU check + A/B timing of the 256 x 256 GEMM tile (csrc/gemm3.hip) through the C ABI.
//   part 1  every mode / epilogue at ragged shapes against a double-precision host reference (operands are small dyadic
//           rationals, so fp32 accumulation is exact and a single misplaced element is visible), both kernel paths;
//   part 2  race screen: the model's shapes, 256-wide path run repeatedly, outputs compared bitwise with the first run and
//           elementwise with the 128-wide path;
//   part 3  timing on the model's shapes (normal random data), 128-wide path vs 256-wide path, back-to-back launches.
// Build + run: tools/native/run_gemm3_check.sh [correct|time|all]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vbx.h"

#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t r; memcpy(&r, &h, 2); return r; }
static float h2f(uint16_t r) { _Float16 h; memcpy(&h, &r, 2); return (float)h; }
template <class T> static T* dev(const std::vector<T>& v) {
  T* p; HIPCHK(hipMalloc(&p, v.size() * sizeof(T) + 256));
  HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p;
}
template <class T> static T* devfill(size_t n, int byte) { T* p; HIPCHK(hipMalloc(&p, n * sizeof(T) + 256)); HIPCHK(hipMemset(p, byte, n * sizeof(T))); return p; }
template <class T> static std::vector<T> host(const T* p, size_t n) { std::vector<T> v(n); HIPCHK(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return v; }
static double gelu(double x) { return 0.5 * x * (1.0 + erf(x * 0.7071067811865476)); }

static int bad = 0;
static void check(const char* what, double got, double want, double tol, long r, long c) {
  if (!(fabs(got - want) <= tol * (1.0 + fabs(want)))) { if (bad < 24) printf("  %s mismatch at (%ld,%ld): got %g want %g\n", what, r, c, got, want); bad++; }
}
static int run(const vbx_gemm_desc& d, const char* what) {
  const int rc = vbx_gemm(&d, nullptr);
  if (rc) { printf("vbx_gemm %s: rc %d %s\n", what, rc, vbx_last_error()); bad++; }
  HIPCHK(hipDeviceSynchronize());
  return rc;
}

// operands with exactly representable values; layout [rows][cols] row-major
struct Mat { int rows, cols; std::vector<float> f; std::vector<uint16_t> b, h; uint16_t *db = nullptr, *dh = nullptr; };
static Mat mk(int rows, int cols, int denom) {
  Mat m{rows, cols};
  m.f.resize((size_t)rows * cols); m.b.resize(m.f.size()); m.h.resize(m.f.size());
  for (size_t i = 0; i < m.f.size(); i++) { m.f[i] = (rand() % 17 - 8) / (float)denom; m.b[i] = f2bf(m.f[i]); m.h[i] = f2h(m.f[i]); }
  m.db = dev(m.b); m.dh = dev(m.h);
  return m;
}

// C[r][c] = sum_k A(r,k) B(c,k) with A given as [M][K] (at = 0) or [K][M] (at = 1), B as [N][K] (bt = 0) or [K][N] (bt = 1)
static std::vector<double> refmm(const Mat& A, int at, const Mat& B, int bt, int M, int N, int K) {
  std::vector<double> C((size_t)M * N);
  for (int r = 0; r < M; r++)
    for (int c = 0; c < N; c++) {
      double s = 0;
      for (int k = 0; k < K; k++) s += (double)(at ? A.f[(size_t)k * M + r] : A.f[(size_t)r * K + k]) * (bt ? B.f[(size_t)k * N + c] : B.f[(size_t)c * K + k]);
      C[(size_t)r * N + c] = s;
    }
  return C;
}

static void correctness(int path) {
  printf("== correctness, path %d (%s)\n", path, path == 2 ? "256x256 gemm3" : (path == 3 ? "128x256 gemm4" : "128-wide"));
  vbx_gemm_select(path);
  srand(7);
  {  // ---- NT, F32 (+bias +resid, bf16 copy), bf16 and fp16 operands; ragged M / N / K
    const int M = 600, N = 392, K = 200;
    Mat A = mk(M, K, 8), B = mk(N, K, 16);
    std::vector<float> bias(N), resid((size_t)M * N);
    for (auto& v : bias) v = (rand() % 9 - 4) / 4.0f;
    for (auto& v : resid) v = (rand() % 33 - 16) / 8.0f;
    float *dbias = dev(bias), *dresid = dev(resid);
    auto C = refmm(A, 0, B, 0, M, N, K);
    for (int f16 = 0; f16 < 2; f16++) {
      float* dC = devfill<float>((size_t)M * N, 0xff); uint16_t* dC2 = devfill<uint16_t>((size_t)M * N, 0xff);
      vbx_gemm_desc d{};
      d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_F32; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N;
      d.A = f16 ? A.dh : A.db; d.B = f16 ? B.dh : B.db; d.C = dC; d.bias = dbias; d.resid = dresid; d.C2 = dC2; d.f16 = f16;
      run(d, "NT F32");
      auto out = host(dC, (size_t)M * N); auto out2 = host(dC2, (size_t)M * N);
      for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) {
        const double w = C[(size_t)r * N + c] + bias[c] + resid[(size_t)r * N + c];
        check(f16 ? "NT.F32.f16" : "NT.F32.bf16", out[(size_t)r * N + c], w, 1e-6, r, c);
        check("NT.F32.copy", bf2f(out2[(size_t)r * N + c]), w, 8e-3, r, c);
      }
      hipFree(dC); hipFree(dC2);
    }
    {  // BF16 output with bias
      uint16_t* dC = devfill<uint16_t>((size_t)M * N, 0xff);
      vbx_gemm_desc d{};
      d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_BF16; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N;
      d.A = A.db; d.B = B.db; d.C = dC; d.bias = dbias;
      run(d, "NT BF16");
      auto out = host(dC, (size_t)M * N);
      for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("NT.BF16", bf2f(out[(size_t)r * N + c]), C[(size_t)r * N + c] + bias[c], 8e-3, r, c);
      hipFree(dC);
    }
  }
  {  // ---- NN (B is [K][N]), BF16 out; and F32 out
    const int M = 520, N = 776, K = 328;
    Mat A = mk(M, K, 8), B = mk(K, N, 16);
    auto C = refmm(A, 0, B, 1, M, N, K);
    uint16_t* dC = devfill<uint16_t>((size_t)M * N, 0xff); float* dF = devfill<float>((size_t)M * N, 0xff);
    vbx_gemm_desc d{};
    d.mode = VBX_GEMM_NN; d.epilogue = VBX_EPI_BF16; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = N; d.ldc = N; d.A = A.db; d.B = B.db; d.C = dC;
    run(d, "NN BF16");
    auto out = host(dC, (size_t)M * N);
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("NN.BF16", bf2f(out[(size_t)r * N + c]), C[(size_t)r * N + c], 8e-3, r, c);
    d.epilogue = VBX_EPI_F32; d.C = dF;
    run(d, "NN F32");
    auto outf = host(dF, (size_t)M * N);
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("NN.F32", outf[(size_t)r * N + c], C[(size_t)r * N + c], 1e-6, r, c);
  }
  {  // ---- NT GEGLU (fp16 operands; fp16 G + bf16 copy + bf16 H1)
    const int M = 300, N = 384, K = 128;
    Mat A = mk(M, K, 8), B = mk(N, K, 16);
    std::vector<float> bias(N);
    for (auto& v : bias) v = (rand() % 9 - 4) / 4.0f;
    float* dbias = dev(bias);
    auto C = refmm(A, 0, B, 0, M, N, K);
    uint16_t *dG = devfill<uint16_t>((size_t)M * N / 2, 0xff), *dGb = devfill<uint16_t>((size_t)M * N / 2, 0xff), *dH1 = devfill<uint16_t>((size_t)M * N, 0xff);
    vbx_gemm_desc d{};
    d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_GEGLU; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N / 2;
    d.A = A.dh; d.B = B.dh; d.C = dG; d.bias = dbias; d.C2 = dH1; d.C3 = dGb; d.f16 = 1;
    run(d, "NT GEGLU");
    auto G = host(dG, (size_t)M * N / 2), Gb = host(dGb, (size_t)M * N / 2), H1 = host(dH1, (size_t)M * N);
    for (int r = 0; r < M; r++) for (int t = 0; t < N / 128; t++) for (int c = 0; c < 64; c++) {
      const double x = C[(size_t)r * N + t * 128 + c] + bias[t * 128 + c], g = C[(size_t)r * N + t * 128 + 64 + c] + bias[t * 128 + 64 + c];
      const double w = gelu(g) * x;
      check("GEGLU.G", h2f(G[(size_t)r * (N / 2) + t * 64 + c]), w, 4e-3, r, t * 64 + c);
      check("GEGLU.Gb", bf2f(Gb[(size_t)r * (N / 2) + t * 64 + c]), w, 1e-2, r, t * 64 + c);
    }
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("GEGLU.H1", bf2f(H1[(size_t)r * N + c]), C[(size_t)r * N + c] + bias[c], 8e-3, r, c);
  }
  {  // ---- NT QKV: head split + MultiheadRMSNorm + rotary
    const int Bb = 3, Np = 100, H = 4, I = H * 64, M = Bb * Np, N = 3 * I, K = 64;
    Mat A = mk(M, K, 8), B = mk(N, K, 16);
    std::vector<float> qg(I), kg(I), rc((size_t)Np * 32), rs((size_t)Np * 32);
    for (auto& v : qg) v = 1.0f + (rand() % 9 - 4) / 16.0f;
    for (auto& v : kg) v = 1.0f + (rand() % 9 - 4) / 16.0f;
    for (int n = 0; n < Np; n++) for (int dd = 0; dd < 32; dd++) { const double ang = (n - 16) * pow(50000.0, -dd / 32.0); rc[n * 32 + dd] = (float)cos(ang); rs[n * 32 + dd] = (float)sin(ang); }
    float *dqg = dev(qg), *dkg = dev(kg), *drc = dev(rc), *drs = dev(rs);
    auto C = refmm(A, 0, B, 0, M, N, K);
    const size_t hs = (size_t)Bb * H * Np * 64;
    uint16_t *q16 = devfill<uint16_t>(hs, 0xff), *k16 = devfill<uint16_t>(hs, 0xff), *qb = devfill<uint16_t>(hs, 0xff), *kb = devfill<uint16_t>(hs, 0xff),
             *v = devfill<uint16_t>(hs, 0xff), *v16 = devfill<uint16_t>(hs, 0xff);
    float *qrn = devfill<float>((size_t)Bb * H * Np, 0xff), *krn = devfill<float>((size_t)Bb * H * Np, 0xff);
    vbx_gemm_desc d{};
    d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_QKV; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.A = A.dh; d.B = B.dh; d.f16 = 1;
    d.Np = Np; d.H = H; d.qk_scale = 8.0f; d.q_gamma = dqg; d.k_gamma = dkg; d.rot_cos = drc; d.rot_sin = drs;
    d.q16 = q16; d.k16 = k16; d.qb = qb; d.kb = kb; d.v = v; d.v16 = v16; d.q_rnorm = qrn; d.k_rnorm = krn;
    run(d, "NT QKV");
    auto hq = host(q16, hs), hk = host(k16, hs), hqb = host(qb, hs), hkb = host(kb, hs), hv = host(v, hs), hv16 = host(v16, hs);
    auto hqrn = host(qrn, (size_t)Bb * H * Np), hkrn = host(krn, (size_t)Bb * H * Np);
    for (int r = 0; r < M; r++) {
      const int b = r / Np, n = r % Np;
      for (int which = 0; which < 3; which++) for (int h = 0; h < H; h++) {
        const double* t = &C[(size_t)r * N + which * I + h * 64];
        const size_t o = (((size_t)b * H + h) * Np + n) * 64;
        if (which == 2) { for (int dd = 0; dd < 64; dd++) { check("QKV.v", bf2f(hv[o + dd]), t[dd], 8e-3, r, dd); check("QKV.v16", h2f(hv16[o + dd]), t[dd], 2e-3, r, dd); } continue; }
        double ss = 0; for (int dd = 0; dd < 64; dd++) ss += t[dd] * t[dd];
        const double rinv = 1.0 / fmax(sqrt(ss), 1e-12);
        const float* gam = which == 0 ? qg.data() : kg.data();
        double u[64], out[64];
        for (int dd = 0; dd < 64; dd++) u[dd] = t[dd] * rinv * 8.0 * gam[h * 64 + dd];
        for (int dd = 0; dd < 32; dd++) { out[dd] = u[dd] * rc[n * 32 + dd] - u[dd + 32] * rs[n * 32 + dd]; out[dd + 32] = u[dd + 32] * rc[n * 32 + dd] + u[dd] * rs[n * 32 + dd]; }
        const auto& h16 = which == 0 ? hq : hk; const auto& hb = which == 0 ? hqb : hkb;
        for (int dd = 0; dd < 64; dd++) { check("QKV.qk16", h2f(h16[o + dd]), out[dd], 2e-3, r, which * 1000 + h * 64 + dd); check("QKV.qkb", bf2f(hb[o + dd]), out[dd], 1e-2, r, dd); }
        check("QKV.rnorm", (which == 0 ? hqrn : hkrn)[((size_t)b * H + h) * Np + n], rinv, 1e-5, r, h);
      }
    }
  }
  {  // ---- TN split-K slabs, single and grouped
    const int Ms[3] = {520, 264, 256}, Ns[3] = {264, 520, 512}, K = 1000, splits = 3;
    vbx_gemm_desc ds[3];
    std::vector<std::vector<double>> refs;
    std::vector<float*> slabs;
    for (int j = 0; j < 3; j++) {
      Mat A = mk(K, Ms[j], 8), B = mk(K, Ns[j], 16);
      refs.push_back(refmm(A, 1, B, 1, Ms[j], Ns[j], K));
      float* dS = devfill<float>((size_t)splits * Ms[j] * Ns[j], 0xff);
      slabs.push_back(dS);
      vbx_gemm_desc d{};
      d.mode = VBX_GEMM_TN; d.epilogue = VBX_EPI_SPLITK; d.M = Ms[j]; d.N = Ns[j]; d.K = K; d.lda = Ms[j]; d.ldb = Ns[j]; d.A = A.db; d.B = B.db; d.C = dS; d.splits = splits;
      ds[j] = d;
    }
    for (int grouped = 0; grouped < 2; grouped++) {
      for (int j = 0; j < 3; j++) HIPCHK(hipMemset(slabs[j], 0xff, (size_t)splits * Ms[j] * Ns[j] * 4));
      if (grouped) {
        const int rc = vbx_gemm_tn_splitk_grouped(ds, 3, nullptr);
        if (rc) { printf("grouped: rc %d %s\n", rc, vbx_last_error()); bad++; }
        HIPCHK(hipDeviceSynchronize());
      } else {
        for (int j = 0; j < 3; j++) run(ds[j], "TN SPLITK");
      }
      for (int j = 0; j < 3; j++) {
        auto s = host(slabs[j], (size_t)splits * Ms[j] * Ns[j]);
        for (int r = 0; r < Ms[j]; r++) for (int c = 0; c < Ns[j]; c++) {
          double t = 0; for (int k = 0; k < splits; k++) t += s[((size_t)k * Ms[j] + r) * Ns[j] + c];
          check(grouped ? "TN.grouped" : "TN.splitk", t, refs[j][(size_t)r * Ns[j] + c], 1e-6, r, c);
        }
      }
    }
  }
  printf("   -> %d mismatches so far\n", bad);
}

// ------------------------------------------------------------------------------------------------ model shapes
static std::vector<uint16_t> randn16(size_t n, float std, bool f16) {
  std::vector<uint16_t> v(n);
  for (size_t i = 0; i < n; i++) {
    float a = 0; for (int k = 0; k < 4; k++) a += (rand() / (float)RAND_MAX - 0.5f);  // ~N(0, 1/3)
    a *= std * 1.732f;
    v[i] = f16 ? f2h(a) : f2bf(a);
  }
  return v;
}
struct Bench {
  std::string name; vbx_gemm_desc d; double flops; std::vector<std::pair<void*, size_t>> outs;
};
static float time_desc(const vbx_gemm_desc& d, int iters) {
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm: %s\n", vbx_last_error()); exit(2); }
  HIPCHK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; i++) vbx_gemm(&d, nullptr);
  HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

static void model_shapes(bool do_time, bool do_race) {
  srand(11);
  const int Bt = 8, Np = 1040, M = Bt * Np, D = 512, H = 16, I = 1024, Fp = 1408;
  auto A512h = dev(randn16((size_t)M * D, 1.0f, true));      // normed activations, fp16
  auto A512b = dev(randn16((size_t)M * D, 1.0f, false));     // bf16 gradient-like
  auto A1024h = dev(randn16((size_t)M * I, 1.0f, true));
  auto A1408h = dev(randn16((size_t)M * Fp, 1.0f, true));
  auto A2816b = dev(randn16((size_t)M * 2 * Fp, 1.0f, false));
  auto A3072b = dev(randn16((size_t)M * 3 * I, 1.0f, false));
  auto A1024b = dev(randn16((size_t)M * I, 1.0f, false));
  auto A1408b = dev(randn16((size_t)M * Fp, 1.0f, false));
  auto Wqkvh = dev(randn16((size_t)3 * I * D, 0.044f, true)), Wqkvb = dev(randn16((size_t)3 * I * D, 0.044f, false));
  auto W1h = dev(randn16((size_t)2 * Fp * D, 0.044f, true)), W1b = dev(randn16((size_t)2 * Fp * D, 0.044f, false));
  auto Wouth = dev(randn16((size_t)D * I, 0.03f, true)), Woutb = dev(randn16((size_t)D * I, 0.03f, false));
  auto W2h = dev(randn16((size_t)D * Fp, 0.027f, true)), W2b = dev(randn16((size_t)D * Fp, 0.027f, false));
  std::vector<float> fb(4096, 0.01f), tab((size_t)Np * 32, 0.7f), gam(I, 1.0f);
  float *bias = dev(fb), *rc = dev(tab), *rs = dev(tab), *qg = dev(gam), *kg = dev(gam);
  float* resid = devfill<float>((size_t)M * D, 0);
  const size_t hs = (size_t)Bt * H * Np * 64;
  uint16_t *q16 = devfill<uint16_t>(hs, 0), *k16 = devfill<uint16_t>(hs, 0), *qb = devfill<uint16_t>(hs, 0), *kb = devfill<uint16_t>(hs, 0), *v = devfill<uint16_t>(hs, 0),
           *v16 = devfill<uint16_t>(hs, 0);
  float *qrn = devfill<float>((size_t)Bt * H * Np, 0), *krn = devfill<float>((size_t)Bt * H * Np, 0);
  uint16_t *G = devfill<uint16_t>((size_t)M * Fp, 0), *Gb = devfill<uint16_t>((size_t)M * Fp, 0), *H1 = devfill<uint16_t>((size_t)M * 2 * Fp, 0);
  float* Cf = devfill<float>((size_t)M * D, 0);
  uint16_t* Cb = devfill<uint16_t>((size_t)M * 3 * I, 0);
  std::vector<Bench> bs;
  auto base = [&](int mode, int epi, int N, int K, const void* A, int lda, const void* B, int ldb) {
    vbx_gemm_desc d{}; d.mode = mode; d.epilogue = epi; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.A = A; d.B = B; return d;
  };
  {
    vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_QKV, 3 * I, D, A512h, D, Wqkvh, D);
    d.f16 = 1; d.Np = Np; d.H = H; d.qk_scale = 8.f; d.q_gamma = qg; d.k_gamma = kg; d.rot_cos = rc; d.rot_sin = rs;
    d.q16 = q16; d.k16 = k16; d.qb = qb; d.kb = kb; d.v = v; d.v16 = v16; d.q_rnorm = qrn; d.k_rnorm = krn;
    bs.push_back({"to_qkv train (NT f16, N=3072 K=512, QKV epi, all copies)", d, 2.0 * M * 3 * I * D, {{q16, hs * 2}, {k16, hs * 2}, {qb, hs * 2}, {v, hs * 2}, {v16, hs * 2}, {qrn, (size_t)Bt * H * Np * 4}}});
    d.qb = d.kb = d.v = nullptr; d.q_rnorm = d.k_rnorm = nullptr;
    bs.push_back({"to_qkv eval  (fp16 outputs only)", d, 2.0 * M * 3 * I * D, {{q16, hs * 2}, {k16, hs * 2}, {v16, hs * 2}}});
  }
  {
    vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_GEGLU, 2 * Fp, D, A512h, D, W1h, D);
    d.f16 = 1; d.C = G; d.ldc = Fp; d.bias = bias; d.C2 = H1; d.C3 = Gb;
    bs.push_back({"ff_in train (NT f16, N=2816 K=512, GEGLU + H1 + bf16 copy)", d, 2.0 * M * 2 * Fp * D, {{G, (size_t)M * Fp * 2}, {Gb, (size_t)M * Fp * 2}, {H1, (size_t)M * 2 * Fp * 2}}});
    d.C2 = nullptr; d.C3 = nullptr;
    bs.push_back({"ff_in eval", d, 2.0 * M * 2 * Fp * D, {{G, (size_t)M * Fp * 2}}});
  }
  { vbx_gemm_desc d = base(VBX_GEMM_NN, VBX_EPI_BF16, Fp, D, A512b, D, W2b, Fp); d.C = Cb; d.ldc = Fp;
    bs.push_back({"dgrad ff_out -> dg (NN bf16, N=1408 K=512)", d, 2.0 * M * Fp * D, {{Cb, (size_t)M * Fp * 2}}}); }
  { vbx_gemm_desc d = base(VBX_GEMM_NN, VBX_EPI_BF16, I, D, A512b, D, Woutb, I); d.C = Cb; d.ldc = I;
    bs.push_back({"dgrad to_out -> dO (NN bf16, N=1024 K=512)", d, 2.0 * M * I * D, {{Cb, (size_t)M * I * 2}}}); }
  { vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_F32, D, I, A1024h, I, Wouth, I); d.f16 = 1; d.C = Cf; d.ldc = D; d.resid = resid;
    bs.push_back({"to_out (NT f16, N=512 K=1024, + resid)", d, 2.0 * M * D * I, {{Cf, (size_t)M * D * 4}}}); }
  { vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_F32, D, Fp, A1408h, Fp, W2h, Fp); d.f16 = 1; d.C = Cf; d.ldc = D; d.resid = resid; d.bias = bias;
    bs.push_back({"ff_out (NT f16, N=512 K=1408, + bias + resid)", d, 2.0 * M * D * Fp, {{Cf, (size_t)M * D * 4}}}); }
  if (do_time && !do_race) {  // what the fp32 residual read costs these launches: the same GEMMs without it
    { vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_F32, D, I, A1024h, I, Wouth, I); d.f16 = 1; d.C = Cf; d.ldc = D;
      bs.push_back({"to_out WITHOUT the residual (timing only)", d, 2.0 * M * D * I, {{Cf, (size_t)M * D * 4}}}); }
    { vbx_gemm_desc d = base(VBX_GEMM_NT, VBX_EPI_BF16, D, I, A1024h, I, Wouth, I); d.f16 = 0; d.A = A1024b; d.B = Woutb; d.C = Cb; d.ldc = D;
      bs.push_back({"to_out shape, bf16 output, no residual (timing only)", d, 2.0 * M * D * I, {{Cb, (size_t)M * D * 2}}}); }
  }
  { vbx_gemm_desc d = base(VBX_GEMM_NN, VBX_EPI_BF16, D, 3 * I, A3072b, 3 * I, Wqkvb, D); d.C = Cb; d.ldc = D;
    bs.push_back({"dgrad to_qkv (NN bf16, N=512 K=3072)", d, 2.0 * M * D * 3 * I, {{Cb, (size_t)M * D * 2}}}); }
  { vbx_gemm_desc d = base(VBX_GEMM_NN, VBX_EPI_BF16, D, 2 * Fp, A2816b, 2 * Fp, W1b, D); d.C = Cb; d.ldc = D;
    bs.push_back({"dgrad ff_in (NN bf16, N=512 K=2816)", d, 2.0 * M * D * 2 * Fp, {{Cb, (size_t)M * D * 2}}}); }

  if (do_race) {
    printf("== race screen / path agreement on the model's shapes\n");
    for (auto& b : bs) {
      std::vector<std::vector<uint8_t>> first, oldp;
      vbx_gemm_select(1); run(b.d, b.name.c_str());
      for (auto& o : b.outs) oldp.push_back(host((const uint8_t*)o.first, o.second));
      for (int newpath = 2; newpath <= 3; newpath++) {
      first.clear();
      vbx_gemm_select(newpath);
      int diffs_run = 0;
      for (int it = 0; it < 6; it++) {
        for (auto& o : b.outs) HIPCHK(hipMemset(o.first, 0xff, o.second));
        run(b.d, b.name.c_str());
        for (size_t k = 0; k < b.outs.size(); k++) {
          auto cur = host((const uint8_t*)b.outs[k].first, b.outs[k].second);
          if (it == 0) first.push_back(cur);
          else if (memcmp(cur.data(), first[k].data(), cur.size())) diffs_run++;
        }
      }
      // old vs new: identical operands, fp32 accumulation in a different order -> compare as 16-bit / fp32 values with a tolerance
      long nd = 0; double worst = 0;
      for (size_t k = 0; k < b.outs.size(); k++) {
        const bool is_f32 = (b.d.epilogue == VBX_EPI_F32) || (b.d.epilogue == VBX_EPI_QKV && b.outs[k].first == (void*)qrn);
        const size_t n = b.outs[k].second / (is_f32 ? 4 : 2);
        for (size_t i = 0; i < n; i++) {
          double x, y;
          if (is_f32) { x = ((const float*)first[k].data())[i]; y = ((const float*)oldp[k].data())[i]; }
          else {
            const uint16_t a = ((const uint16_t*)first[k].data())[i], c = ((const uint16_t*)oldp[k].data())[i];
            const bool f16out = (b.d.epilogue == VBX_EPI_QKV && (b.outs[k].first == (void*)q16 || b.outs[k].first == (void*)k16 || b.outs[k].first == (void*)v16)) ||
                                (b.d.epilogue == VBX_EPI_GEGLU && b.outs[k].first == (void*)G);
            x = f16out ? h2f(a) : bf2f(a); y = f16out ? h2f(c) : bf2f(c);
          }
          const double e = fabs(x - y) / (1.0 + fabs(y));
          if (!(e <= 2e-2)) nd++;
          if (e > worst || e != e) worst = e;
        }
      }
      printf("  %-62s path %d reruns differing: %d   |new-old| > 2e-2: %ld (worst %.3g)\n", b.name.c_str(), newpath, diffs_run, nd, worst);
      if (diffs_run || nd) bad++;
      }
    }
  }
  if (do_time) {
    printf("== timing (us per launch, back to back, normal random data)\n");
    for (auto& b : bs) {
      float t[3][2];
      for (int rep = 0; rep < 2; rep++)
        for (int path = 1; path <= 3; path++) { vbx_gemm_select(path); t[path - 1][rep] = time_desc(b.d, 20); }
      const float t1 = fminf(t[0][0], t[0][1]), t2 = fminf(t[1][0], t[1][1]), t3 = fminf(t[2][0], t[2][1]);
      printf("  %-62s 128-wide %6.1f us (%5.0f TF/s)  gemm3 %6.1f us (%5.0f)  gemm4 %6.1f us (%5.0f TF/s)\n", b.name.c_str(), t1, b.flops / t1 * 1e-6, t2,
             b.flops / t2 * 1e-6, t3, b.flops / t3 * 1e-6);
    }
    // weight gradients of a layer: four launches (128-wide, their own split counts) vs one grouped launch
    {
      float* slab = devfill<float>((size_t)16 * 3 * I * D, 0);
      vbx_gemm_desc w[4];
      auto tn = [&](const void* P, int I_, const void* Q, int J_, int splits) {
        vbx_gemm_desc d{}; d.mode = VBX_GEMM_TN; d.epilogue = VBX_EPI_SPLITK; d.M = I_; d.N = J_; d.K = M; d.lda = I_; d.ldb = J_; d.A = P; d.B = Q; d.C = slab; d.splits = splits; return d;
      };
      const double fl = 2.0 * M * ((double)3 * I * D + (double)D * I + (double)2 * Fp * D + (double)D * Fp);
      for (int s3 = 2; s3 <= 4; s3++) {
        w[0] = tn(A3072b, 3 * I, A512b, D, s3); w[1] = tn(A512b, D, A1024b, I, s3); w[2] = tn(A2816b, 2 * Fp, A512b, D, s3); w[3] = tn(A512b, D, A1408b, Fp, s3);
        // (all four share one slab buffer here: timing only)
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        vbx_gemm_select(2);
        for (int i = 0; i < 3; i++) vbx_gemm_tn_splitk_grouped(w, 4, nullptr);
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 20; i++) vbx_gemm_tn_splitk_grouped(w, 4, nullptr);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        printf("  layer wgrads, ONE grouped 256-wide launch, %d splits: %7.1f us (%6.0f TF/s)\n", s3, ms * 50.f, fl / (ms * 50.f) * 1e-6);
      }
      vbx_gemm_select(1);
      const int so[4] = {4, 12, 4, 8};
      w[0] = tn(A3072b, 3 * I, A512b, D, so[0]); w[1] = tn(A512b, D, A1024b, I, so[1]); w[2] = tn(A2816b, 2 * Fp, A512b, D, so[2]); w[3] = tn(A512b, D, A1408b, Fp, so[3]);
      float tot = 0; for (int j = 0; j < 4; j++) tot += time_desc(w[j], 20);
      printf("  layer wgrads, four 128-wide launches (splits 4/12/4/8): %7.1f us (%6.0f TF/s)\n", tot, fl / tot * 1e-6);
    }
    // K sweep at the to_qkv output shape with a plain bf16 epilogue: slope = k-loop rate, intercept = prologue + epilogue + launch
    {
      const int N = 3072, Kmax = 4096;
      auto X = dev(randn16((size_t)M * Kmax, 1.0f, false)); auto Y = dev(randn16((size_t)N * Kmax, 0.03f, false)); uint16_t* Z = devfill<uint16_t>((size_t)M * N, 0);
      for (int K : {128, 256, 512, 1024, 2048, 4096}) {
        vbx_gemm_desc d{}; d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_BF16; d.M = M; d.N = N; d.K = K; d.lda = Kmax; d.ldb = Kmax; d.ldc = N; d.A = X; d.B = Y; d.C = Z;
        float t[3];
        for (int path = 1; path <= 3; path++) { vbx_gemm_select(path); t[path - 1] = fminf(time_desc(d, 20), time_desc(d, 20)); }
        printf("  K sweep M=8320 N=3072 K=%4d (NT bf16 -> bf16): 128-wide %6.1f us  gemm3 %6.1f us  gemm4 %6.1f us   (%.0f / %.0f / %.0f TF/s)\n", K, t[0], t[1], t[2],
               2.0 * M * N * K / t[0] * 1e-6, 2.0 * M * N * K / t[1] * 1e-6, 2.0 * M * N * K / t[2] * 1e-6);
      }
      hipFree(X); hipFree(Y); hipFree(Z);
    }
    // square reference points (NT bf16 -> bf16)
    for (int n : {4096, 8192}) {
      auto X = dev(randn16((size_t)n * n, 1.0f, false)); auto Y = dev(randn16((size_t)n * n, 1.0f, false)); uint16_t* Z = devfill<uint16_t>((size_t)n * n, 0);
      vbx_gemm_desc d{}; d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_BF16; d.M = d.N = d.K = n; d.lda = d.ldb = d.ldc = n; d.A = X; d.B = Y; d.C = Z;
      for (int path = 1; path <= 3; path++) { vbx_gemm_select(path); const float t = time_desc(d, 10); printf("  %d^3 NT bf16 path %d: %8.1f us  %6.0f TF/s\n", n, path, t, 2.0 * n * n * n / t * 1e-6); }
      hipFree(X); hipFree(Y); hipFree(Z);
    }
  }
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "all";
  if (what == "correct" || what == "all") { correctness(1); correctness(2); correctness(3); }
  if (what == "race" || what == "all") model_shapes(false, true);
  if (what == "time" || what == "all") model_shapes(true, false);
  printf(bad ? "GEMM3 CHECK FAILED: %d problems\n" : "GEMM3 CHECK OK\n", bad);
  return bad != 0;
}
