// Stand-alone (no Python / torch) GPU check of the GEMM epilogues through the C ABI: C = A . B^T (NT, bf16 / fp16 operands) with
//  * VBX_EPI_F32  : + bias + residual, fp32 out and bf16 copy         (FeedForward-out / to_out / to_embed)
//  * VBX_EPI_BF16 : + bias, bf16 out
//  * VBX_EPI_GEGLU: gated GELU over the packed [64 x | 64 gate] column tiles, fp16 G, bf16 copy, bf16 pre-activation H1
// against a double-precision host reference, at a ragged M (partial last tile).  Build + run: tools/native/run_epi_check.sh
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vbx.h"

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t r; memcpy(&r, &h, 2); return r; }
static float h2f(uint16_t r) { _Float16 h; memcpy(&h, &r, 2); return (float)h; }
template <class T> static T* dev(const std::vector<T>& v) {
  T* p; if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) { printf("hipMalloc failed\n"); exit(2); }
  hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return p;
}
template <class T> static std::vector<T> host(const T* p, size_t n) { std::vector<T> v(n); hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost); return v; }
static double gelu(double x) { return 0.5 * x * (1.0 + erf(x * 0.7071067811865476)); }

int main() {
  const int M = 200, N = 256, K = 64;
  srand(1);
  std::vector<float> A(M * K), B(N * K), bias(N), resid((size_t)M * N);
  for (auto& v : A) v = (rand() % 17 - 8) / 8.0f;
  for (auto& v : B) v = (rand() % 17 - 8) / 16.0f;
  for (auto& v : bias) v = (rand() % 9 - 4) / 4.0f;
  for (auto& v : resid) v = (rand() % 33 - 16) / 8.0f;
  std::vector<uint16_t> Ab(M * K), Bb(N * K), Ah(M * K), Bh(N * K);
  for (int i = 0; i < M * K; i++) { Ab[i] = f2bf(A[i]); Ah[i] = f2h(A[i]); }   // exactly representable in both formats
  for (int i = 0; i < N * K; i++) { Bb[i] = f2bf(B[i]); Bh[i] = f2h(B[i]); }
  std::vector<double> C((size_t)M * N);
  for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) { double s = 0; for (int k = 0; k < K; k++) s += (double)A[r * K + k] * B[c * K + k]; C[(size_t)r * N + c] = s; }
  uint16_t *dAb = dev(Ab), *dBb = dev(Bb), *dAh = dev(Ah), *dBh = dev(Bh);
  float *dbias = dev(bias), *dresid = dev(resid);
  int bad = 0;
  auto check = [&](const char* what, double got, double want, double tol, int r, int c) {
    if (!(fabs(got - want) <= tol * (1.0 + fabs(want)))) { if (bad < 8) printf("%s mismatch at (%d,%d): got %g want %g\n", what, r, c, got, want); bad++; }
  };
  vbx_gemm_desc d{};
  d.mode = VBX_GEMM_NT; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N; d.bias = dbias;
  {  // ---- F32 (+ bias + residual, bf16 copy), bf16 operands
    std::vector<float> out((size_t)M * N, -777.f); std::vector<uint16_t> out2((size_t)M * N, 0xdead);
    float* dC = dev(out); uint16_t* dC2 = dev(out2);
    d.epilogue = VBX_EPI_F32; d.A = dAb; d.B = dBb; d.C = dC; d.resid = dresid; d.C2 = dC2; d.f16 = 0;
    if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm F32: %s\n", vbx_last_error()); return 2; }
    hipDeviceSynchronize();
    out = host(dC, out.size()); out2 = host(dC2, out2.size());
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) {
      const double w = C[(size_t)r * N + c] + bias[c] + resid[(size_t)r * N + c];
      check("F32", out[(size_t)r * N + c], w, 1e-5, r, c); check("F32.bf16copy", bf2f(out2[(size_t)r * N + c]), w, 8e-3, r, c);
    }
  }
  {  // ---- BF16 (+ bias)
    std::vector<uint16_t> out((size_t)M * N, 0xdead); uint16_t* dC = dev(out);
    d.epilogue = VBX_EPI_BF16; d.A = dAb; d.B = dBb; d.C = dC; d.resid = nullptr; d.C2 = nullptr; d.f16 = 0;
    if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm BF16: %s\n", vbx_last_error()); return 2; }
    hipDeviceSynchronize();
    out = host(dC, out.size());
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("BF16", bf2f(out[(size_t)r * N + c]), C[(size_t)r * N + c] + bias[c], 8e-3, r, c);
  }
  {  // ---- GEGLU (fp16 operands, fp16 G + bf16 copy + bf16 H1)
    std::vector<uint16_t> G((size_t)M * N / 2, 0xdead), Gb = G, H1((size_t)M * N, 0xdead);
    uint16_t *dG = dev(G), *dGb = dev(Gb), *dH1 = dev(H1);
    d.epilogue = VBX_EPI_GEGLU; d.A = dAh; d.B = dBh; d.C = dG; d.ldc = N / 2; d.C2 = dH1; d.C3 = dGb; d.f16 = 1; d.resid = nullptr;
    if (vbx_gemm(&d, nullptr)) { printf("vbx_gemm GEGLU: %s\n", vbx_last_error()); return 2; }
    hipDeviceSynchronize();
    G = host(dG, G.size()); Gb = host(dGb, Gb.size()); H1 = host(dH1, H1.size());
    for (int r = 0; r < M; r++) for (int t = 0; t < N / 128; t++) for (int c = 0; c < 64; c++) {
      const double x = C[(size_t)r * N + t * 128 + c] + bias[t * 128 + c], g = C[(size_t)r * N + t * 128 + 64 + c] + bias[t * 128 + 64 + c];
      const double w = gelu(g) * x;
      check("GEGLU.G", h2f(G[(size_t)r * (N / 2) + t * 64 + c]), w, 4e-3, r, t * 64 + c);   // A&S erf approximation + fp16
      check("GEGLU.Gb", bf2f(Gb[(size_t)r * (N / 2) + t * 64 + c]), w, 1e-2, r, t * 64 + c);
    }
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) check("GEGLU.H1", bf2f(H1[(size_t)r * N + c]), C[(size_t)r * N + c] + bias[c], 8e-3, r, c);
  }
  printf(bad ? "EPI CHECK FAILED: %d mismatches\n" : "EPI CHECK OK (%d mismatches)\n", bad);
  return bad != 0;
}
