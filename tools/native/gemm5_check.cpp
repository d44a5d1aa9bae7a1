// Stand-alone (no Python / torch) GPU check + A/B timing of the weight-stationary K = 512 GEMM (csrc/gemm5.hip) through the C ABI.
//   correct  QKV and GEGLU epilogues (training outputs and inference outputs, fp16 and bf16 operands) at ragged row counts, few and
//            many heads, against a double-precision host reference -- operands are small dyadic rationals, so fp32 accumulation is
//            exact and a single misplaced element is visible; rows / columns beyond the outputs are checked untouched (0xff fill);
//   race     the model's shapes run repeatedly: outputs bitwise equal from run to run, and elementwise close to the 128-wide path;
//   time     back-to-back launches on the model's shapes, gemm5 against the 128-wide path (and gemm4 for the inference FeedForward-in).
// Build + run: tools/native/run_gemm5_check.sh [correct|race|time|all]
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
  if (!(fabs(got - want) <= tol * (1.0 + fabs(want)))) { if (bad < 32) printf("  %s mismatch at (%ld,%ld): got %g want %g\n", what, r, c, got, want); bad++; }
}
static int run(const vbx_gemm_desc& d, const char* what) {
  const int rc = vbx_gemm(&d, nullptr);
  if (rc) { printf("vbx_gemm %s: rc %d %s\n", what, rc, vbx_last_error()); bad++; }
  HIPCHK(hipDeviceSynchronize());
  return rc;
}
struct Mat { int rows, cols; std::vector<float> f; std::vector<uint16_t> b, h; uint16_t *db = nullptr, *dh = nullptr; };
static Mat mk(int rows, int cols, int denom) {
  Mat m{rows, cols};
  m.f.resize((size_t)rows * cols); m.b.resize(m.f.size()); m.h.resize(m.f.size());
  for (size_t i = 0; i < m.f.size(); i++) { m.f[i] = (rand() % 17 - 8) / (float)denom; m.b[i] = f2bf(m.f[i]); m.h[i] = f2h(m.f[i]); }
  m.db = dev(m.b); m.dh = dev(m.h);
  return m;
}
static std::vector<double> refmm(const Mat& A, const Mat& B, int M, int N, int K) {  // C = A . B^T
  std::vector<double> C((size_t)M * N);
  for (int r = 0; r < M; r++)
    for (int c = 0; c < N; c++) {
      double s = 0;
      const float* a = &A.f[(size_t)r * K]; const float* b = &B.f[(size_t)c * K];
      for (int k = 0; k < K; k++) s += (double)a[k] * b[k];
      C[(size_t)r * N + c] = s;
    }
  return C;
}
static void untouched(const char* what, const std::vector<uint16_t>& v, size_t from) {
  for (size_t i = from; i < v.size(); i++) if (v[i] != 0xffff) { if (bad < 32) printf("  %s: element %zu past the output was written\n", what, i); bad++; return; }
}

static void qkv_case(int Bb, int Np, int H, int f16, bool train, float qk_scale) {
  const int I = H * 64, M = Bb * Np, N = 3 * I, K = 512;
  printf("  QKV  B=%d Np=%d H=%d %s %s qk_scale=%g\n", Bb, Np, H, f16 ? "fp16" : "bf16", train ? "train" : "eval", qk_scale);
  Mat A = mk(M, K, 8), B = mk(N, K, 16);
  std::vector<float> qg(I), kg(I), rc((size_t)Np * 32), rs((size_t)Np * 32);
  for (auto& v : qg) v = 1.0f + (rand() % 9 - 4) / 16.0f;
  for (auto& v : kg) v = 1.0f + (rand() % 9 - 4) / 16.0f;
  for (int n = 0; n < Np; n++) for (int dd = 0; dd < 32; dd++) { const double ang = (n - 16) * pow(50000.0, -dd / 32.0); rc[n * 32 + dd] = (float)cos(ang); rs[n * 32 + dd] = (float)sin(ang); }
  float *dqg = dev(qg), *dkg = dev(kg), *drc = dev(rc), *drs = dev(rs);
  auto C = refmm(A, B, M, N, K);
  const size_t hs = (size_t)Bb * H * Np * 64, pad = 4096;
  uint16_t *q16 = devfill<uint16_t>(hs + pad, 0xff), *k16 = devfill<uint16_t>(hs + pad, 0xff), *qb = devfill<uint16_t>(hs + pad, 0xff), *kb = devfill<uint16_t>(hs + pad, 0xff),
           *v = devfill<uint16_t>(hs + pad, 0xff), *v16 = devfill<uint16_t>(hs + pad, 0xff);
  float *qrn = devfill<float>((size_t)Bb * H * Np, 0xff), *krn = devfill<float>((size_t)Bb * H * Np, 0xff);
  vbx_gemm_desc d{};
  d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_QKV; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.A = f16 ? A.dh : A.db; d.B = f16 ? B.dh : B.db; d.f16 = f16;
  d.Np = Np; d.H = H; d.qk_scale = qk_scale; d.q_gamma = dqg; d.k_gamma = dkg; d.rot_cos = drc; d.rot_sin = drs; d.q_prescale = 1.4375f;
  d.q16 = q16; d.k16 = k16; d.v16 = v16;
  if (train) { d.qb = qb; d.kb = kb; d.v = v; d.q_rnorm = qrn; d.k_rnorm = krn; }
  run(d, "NT QKV");
  auto hq = host(q16, hs + pad), hk = host(k16, hs + pad), hqb = host(qb, hs + pad), hkb = host(kb, hs + pad), hv = host(v, hs + pad), hv16 = host(v16, hs + pad);
  auto hqrn = host(qrn, (size_t)Bb * H * Np), hkrn = host(krn, (size_t)Bb * H * Np);
  untouched("q16", hq, hs); untouched("k16", hk, hs); untouched("v16", hv16, hs);
  if (train) { untouched("qb", hqb, hs); untouched("kb", hkb, hs); untouched("v", hv, hs); }
  else { untouched("qb (eval)", hqb, 0); untouched("v (eval)", hv, 0); }
  for (int r = 0; r < M; r++) {
    const int b = r / Np, n = r % Np;
    for (int which = 0; which < 3; which++) for (int h = 0; h < H; h++) {
      const double* t = &C[(size_t)r * N + which * I + h * 64];
      const size_t o = (((size_t)b * H + h) * Np + n) * 64;
      if (which == 2) {
        for (int dd = 0; dd < 64; dd++) { if (train) check("QKV.v", bf2f(hv[o + dd]), t[dd], 8e-3, r, dd); check("QKV.v16", h2f(hv16[o + dd]), t[dd], 2e-3, r, dd); }
        continue;
      }
      double ss = 0; for (int dd = 0; dd < 64; dd++) ss += t[dd] * t[dd];
      const double rinv = 1.0 / fmax(sqrt(ss), 1e-12);
      const float* gam = which == 0 ? qg.data() : kg.data();
      double u[64], out[64];
      for (int dd = 0; dd < 64; dd++) u[dd] = qk_scale > 0 ? t[dd] * rinv * qk_scale * gam[h * 64 + dd] : t[dd];
      for (int dd = 0; dd < 32; dd++) { out[dd] = u[dd] * rc[n * 32 + dd] - u[dd + 32] * rs[n * 32 + dd]; out[dd + 32] = u[dd + 32] * rc[n * 32 + dd] + u[dd] * rs[n * 32 + dd]; }
      const auto& h16 = which == 0 ? hq : hk; const auto& hb = which == 0 ? hqb : hkb;
      const double ps = which == 0 ? 1.4375 : 1.0;
      for (int dd = 0; dd < 64; dd++) {
        check(which == 0 ? "QKV.q16" : "QKV.k16", h2f(h16[o + dd]), out[dd] * ps, 2e-3, r, h * 64 + dd);
        if (train) check("QKV.qkb", bf2f(hb[o + dd]), out[dd], 1e-2, r, h * 64 + dd);
      }
      if (train) check("QKV.rnorm", (which == 0 ? hqrn : hkrn)[((size_t)b * H + h) * Np + n], rinv, 1e-5, r, h);
    }
  }
  for (void* p : {(void*)q16, (void*)k16, (void*)qb, (void*)kb, (void*)v, (void*)v16, (void*)qrn, (void*)krn, (void*)dqg, (void*)dkg, (void*)drc, (void*)drs, (void*)A.db, (void*)A.dh, (void*)B.db, (void*)B.dh}) HIPCHK(hipFree(p));
}

static void geglu_case(int M, int N, int f16, bool train) {
  const int K = 512;
  printf("  GEGLU M=%d N=%d %s %s\n", M, N, f16 ? "fp16" : "bf16", train ? "train" : "eval");
  Mat A = mk(M, K, 8), B = mk(N, K, 16);
  std::vector<float> bias(N);
  for (auto& v : bias) v = (rand() % 9 - 4) / 4.0f;
  float* dbias = dev(bias);
  auto C = refmm(A, B, M, N, K);
  const size_t pad = 4096;
  uint16_t *dG = devfill<uint16_t>((size_t)M * N / 2 + pad, 0xff), *dGb = devfill<uint16_t>((size_t)M * N / 2 + pad, 0xff), *dH1 = devfill<uint16_t>((size_t)M * N + pad, 0xff);
  vbx_gemm_desc d{};
  d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_GEGLU; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N / 2;
  d.A = f16 ? A.dh : A.db; d.B = f16 ? B.dh : B.db; d.C = dG; d.bias = dbias; d.f16 = f16;
  if (train) { d.C2 = dH1; d.C3 = dGb; }
  run(d, "NT GEGLU");
  auto G = host(dG, (size_t)M * N / 2 + pad), Gb = host(dGb, (size_t)M * N / 2 + pad), H1 = host(dH1, (size_t)M * N + pad);
  untouched("G", G, (size_t)M * N / 2);
  if (train) { untouched("Gb", Gb, (size_t)M * N / 2); untouched("H1", H1, (size_t)M * N); } else { untouched("Gb (eval)", Gb, 0); untouched("H1 (eval)", H1, 0); }
  for (int r = 0; r < M; r++) for (int t = 0; t < N / 128; t++) for (int c = 0; c < 64; c++) {
    const double x = C[(size_t)r * N + t * 128 + c] + bias[t * 128 + c], g = C[(size_t)r * N + t * 128 + 64 + c] + bias[t * 128 + 64 + c];
    const double w = gelu(g) * x;
    const uint16_t gv = G[(size_t)r * (N / 2) + t * 64 + c];
    check("GEGLU.G", f16 ? h2f(gv) : bf2f(gv), w, f16 ? 4e-3 : 1e-2, r, t * 64 + c);
    if (train) check("GEGLU.Gb", bf2f(Gb[(size_t)r * (N / 2) + t * 64 + c]), w, 1e-2, r, t * 64 + c);
  }
  if (train) for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("GEGLU.H1", bf2f(H1[(size_t)r * N + c]), C[(size_t)r * N + c] + bias[c], 8e-3, r, c);
  for (void* p : {(void*)dG, (void*)dGb, (void*)dH1, (void*)dbias, (void*)A.db, (void*)A.dh, (void*)B.db, (void*)B.dh}) HIPCHK(hipFree(p));
}

static void correctness() {
  printf("== correctness (gemm5 path)\n");
  vbx_gemm_select(4);
  srand(7);
  qkv_case(3, 100, 4, 1, true, 8.0f);     // 300 rows: ragged last block, batches change inside a block, 12 slabs = 3 panels
  qkv_case(2, 77, 2, 1, false, 8.0f);     // 6 slabs: the last panel has two idle waves
  qkv_case(1, 24, 6, 0, true, 8.0f);      // fewer rows than a block, Np < 32, bf16 operands
  qkv_case(5, 1040, 16, 1, true, 8.0f);   // 5200 rows: several blocks per workgroup, the benchmark's heads
  qkv_case(2, 33, 2, 1, true, 0.0f);      // no qk-norm
  geglu_case(300, 384, 1, true);
  geglu_case(70, 128, 0, false);
  geglu_case(2100, 2816, 1, true);
  geglu_case(1234, 2816, 1, false);
}

static std::vector<uint16_t> randn16(size_t n, float scale, bool f16) {
  std::vector<uint16_t> v(n);
  for (size_t i = 0; i < n; i++) {
    float u1 = (rand() + 1.0f) / ((float)RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
    float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * scale;
    v[i] = f16 ? f2h(z) : f2bf(z);
  }
  return v;
}
struct Bench { std::string name; vbx_gemm_desc d; double flops; std::vector<std::pair<void*, size_t>> outs; std::vector<int> is_f16, is_f32; };
static float time_desc(const vbx_gemm_desc& d, int iters) {
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm: %s\n", vbx_last_error()); exit(2); }
  HIPCHK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; i++) vbx_gemm(&d, nullptr);
  HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / iters;
}

static void model_shapes(bool do_time, bool do_race, int Bt) {
  srand(11);
  const int Np = 1040, M = Bt * Np, D = 512, H = 16, I = 1024, Fp = 1408;
  auto A512h = dev(randn16((size_t)M * D, 1.0f, true));
  auto Wqkvh = dev(randn16((size_t)3 * I * D, 0.044f, true));
  auto W1h = dev(randn16((size_t)2 * Fp * D, 0.044f, true));
  std::vector<float> fb(4096, 0.01f), tabc((size_t)Np * 32), tabs((size_t)Np * 32), gam(I, 1.0f);
  for (int n = 0; n < Np; n++) for (int dd = 0; dd < 32; dd++) { const double ang = (n - 16) * pow(50000.0, -dd / 32.0); tabc[n * 32 + dd] = (float)cos(ang); tabs[n * 32 + dd] = (float)sin(ang); }
  float *bias = dev(fb), *rc = dev(tabc), *rs = dev(tabs), *qg = dev(gam), *kg = dev(gam);
  const size_t hs = (size_t)Bt * H * Np * 64;
  uint16_t *q16 = devfill<uint16_t>(hs, 0), *k16 = devfill<uint16_t>(hs, 0), *qb = devfill<uint16_t>(hs, 0), *kb = devfill<uint16_t>(hs, 0), *v = devfill<uint16_t>(hs, 0),
           *v16 = devfill<uint16_t>(hs, 0);
  float *qrn = devfill<float>((size_t)Bt * H * Np, 0), *krn = devfill<float>((size_t)Bt * H * Np, 0);
  uint16_t *G = devfill<uint16_t>((size_t)M * Fp, 0), *Gb = devfill<uint16_t>((size_t)M * Fp, 0), *H1 = devfill<uint16_t>((size_t)M * 2 * Fp, 0);
  std::vector<Bench> bs;
  auto base = [&](int epi, int N, const void* A, const void* B) {
    vbx_gemm_desc d{}; d.mode = VBX_GEMM_NT; d.epilogue = epi; d.M = M; d.N = N; d.K = D; d.lda = D; d.ldb = D; d.A = A; d.B = B; d.f16 = 1; return d;
  };
  {
    vbx_gemm_desc d = base(VBX_EPI_QKV, 3 * I, A512h, Wqkvh);
    d.Np = Np; d.H = H; d.qk_scale = 8.f; d.q_gamma = qg; d.k_gamma = kg; d.rot_cos = rc; d.rot_sin = rs; d.q_prescale = 14.427f;
    d.q16 = q16; d.k16 = k16; d.qb = qb; d.kb = kb; d.v = v; d.v16 = v16; d.q_rnorm = qrn; d.k_rnorm = krn;
    bs.push_back({"to_qkv train (all copies)", d, 2.0 * M * 3 * I * D, {{q16, hs * 2}, {k16, hs * 2}, {v16, hs * 2}, {qb, hs * 2}, {kb, hs * 2}, {v, hs * 2}, {qrn, (size_t)Bt * H * Np * 4}, {krn, (size_t)Bt * H * Np * 4}}, {1, 1, 1, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 1, 1}});
    d.qb = d.kb = d.v = nullptr; d.q_rnorm = d.k_rnorm = nullptr;
    bs.push_back({"to_qkv eval  (fp16 outputs only)", d, 2.0 * M * 3 * I * D, {{q16, hs * 2}, {k16, hs * 2}, {v16, hs * 2}}, {1, 1, 1}, {0, 0, 0}});
  }
  {
    vbx_gemm_desc d = base(VBX_EPI_GEGLU, 2 * Fp, A512h, W1h);
    d.C = G; d.ldc = Fp; d.bias = bias; d.C2 = H1; d.C3 = Gb;
    bs.push_back({"ff_in train (G + H1 + bf16 copy)", d, 2.0 * M * 2 * Fp * D, {{G, (size_t)M * Fp * 2}, {Gb, (size_t)M * Fp * 2}, {H1, (size_t)M * 2 * Fp * 2}}, {1, 0, 0}, {0, 0, 0}});
    d.C2 = nullptr; d.C3 = nullptr;
    bs.push_back({"ff_in eval", d, 2.0 * M * 2 * Fp * D, {{G, (size_t)M * Fp * 2}}, {1}, {0}});
  }
  {  // the K = 512 dgrads as NT products on transposed weight copies (plain bf16 epilogue); the same product as NN for comparison
    auto A512b = dev(randn16((size_t)M * D, 1.0f, false));
    auto WoT = dev(randn16((size_t)I * D, 0.03f, false)), W2T = dev(randn16((size_t)Fp * D, 0.03f, false));
    uint16_t* Cb = devfill<uint16_t>((size_t)M * Fp, 0);
    for (int N : {I, Fp}) {
      vbx_gemm_desc d{}; d.mode = VBX_GEMM_NT; d.epilogue = VBX_EPI_BF16; d.M = M; d.N = N; d.K = D; d.lda = D; d.ldb = D; d.ldc = N;
      d.A = A512b; d.B = N == I ? WoT : W2T; d.C = Cb;
      bs.push_back({N == I ? "dgrad to_out as NT (N=1024)" : "dgrad ff_out as NT (N=1408)", d, 2.0 * M * N * D, {{Cb, (size_t)M * N * 2}}, {0}, {0}});
      vbx_gemm_desc e = d; e.mode = VBX_GEMM_NN; e.ldb = N;  // (timing only: the same buffer read as [512][N])
      if (do_time && !do_race) bs.push_back({N == I ? "dgrad to_out as NN (today)" : "dgrad ff_out as NN (today)", e, 2.0 * M * N * D, {{Cb, (size_t)M * N * 2}}, {0}, {0}});
    }
  }
  if (do_race) {
    printf("== race screen / agreement with the 128-wide path, batch %d\n", Bt);
    for (auto& b : bs) {
      std::vector<std::vector<uint8_t>> first, oldp;
      vbx_gemm_select(1);
      for (auto& o : b.outs) HIPCHK(hipMemset(o.first, 0xff, o.second));
      run(b.d, b.name.c_str());
      for (auto& o : b.outs) oldp.push_back(host((const uint8_t*)o.first, o.second));
      vbx_gemm_select(4);
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
      long nd = 0; double worst = 0;
      for (size_t k = 0; k < b.outs.size(); k++) {
        const size_t n = b.outs[k].second / (b.is_f32[k] ? 4 : 2);
        for (size_t i = 0; i < n; i++) {
          double x, y;
          if (b.is_f32[k]) { x = ((const float*)first[k].data())[i]; y = ((const float*)oldp[k].data())[i]; }
          else {
            const uint16_t a = ((const uint16_t*)first[k].data())[i], c = ((const uint16_t*)oldp[k].data())[i];
            x = b.is_f16[k] ? h2f(a) : bf2f(a); y = b.is_f16[k] ? h2f(c) : bf2f(c);
          }
          const double e = fabs(x - y) / (1.0 + fabs(y));
          if (!(e <= 2e-2)) nd++;
          if (e > worst || e != e) worst = e;
        }
      }
      printf("  %-40s reruns differing: %d   |gemm5 - 128-wide| > 2e-2: %ld (worst %.3g)\n", b.name.c_str(), diffs_run, nd, worst);
      if (diffs_run || nd) bad++;
    }
  }
  if (do_time) {
    printf("== timing, batch %d (us per launch, back to back, normal random data)\n", Bt);
    for (auto& b : bs) {
      float t[3][2];
      const int paths[3] = {1, 3, 4};
      for (int rep = 0; rep < 2; rep++)
        for (int pi = 0; pi < 3; pi++) { vbx_gemm_select(paths[pi]); t[pi][rep] = time_desc(b.d, 20); }
      const float t1 = fminf(t[0][0], t[0][1]), t3 = fminf(t[1][0], t[1][1]), t5 = fminf(t[2][0], t[2][1]);
      printf("  %-40s 128-wide %6.1f us (%5.0f TF/s)  gemm4 %6.1f us (%5.0f)  gemm5 %6.1f us (%5.0f TF/s)\n", b.name.c_str(), t1, b.flops / t1 * 1e-6, t3,
             b.flops / t3 * 1e-6, t5, b.flops / t5 * 1e-6);
    }
  }
  vbx_gemm_select(0);
}

int main(int argc, char** argv) {
  const std::string what = argc > 1 ? argv[1] : "all";
  if (what == "correct" || what == "all") correctness();
  if (what == "race" || what == "all") model_shapes(false, true, 8);
  if (what == "time" || what == "all") { model_shapes(true, false, 8); model_shapes(true, false, 4); }
  printf(bad ? "FAILED: %d mismatches\n" : "ok\n", bad);
  return bad ? 1 : 0;
}
