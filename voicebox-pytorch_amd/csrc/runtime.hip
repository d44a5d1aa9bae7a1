// Stage-level runtime: VoiceBox forward / backward as native launch sequences over caller-owned arenas.
// Mirrors VoiceBox.forward (voicebox_pytorch.py:987-1115) + Transformer.forward (:412-479).
//
// HBM layout (all token-major, row r = b*Np + n, Np = N + R registers in place at rows n < R):
//   residual stream   fp32 [M, D]    one snapshot per sub-layer in training (xs[0..2L]), 2 ping-pong buffers in eval
//   normed inputs     fp16 [M, D]    hn1 / hn2 (forward GEMM A operands)
//   q-hat, k-hat, v   fp16 [B,H,Np,64]
//   attention out     fp16 [M, H*64], log2-LSE fp32 [B,H,Np]
//   FF pre-activation bf16 [M, 2*Fp] (interleaved x|gate per 128 columns, training only), GEGLU out fp16 [M, Fp], Fp = ceil64(F)
// Precision contract: every FORWARD GEMM/attention operand is fp16 (fp32 accumulate): the qk-normed logits
// 10*q.k have std ~80, so a 2^-9 (bf16) relative error anywhere upstream of q/k moves them by ~0.2 and the
// prediction by ~5%; fp16 (2^-11) costs the same MFMA rate and brings that to ~1%.  Every BACKWARD GEMM operand
// is bf16 (gradients need the exponent range), so in training each saved activation also has a bf16 copy.
#include "common.hpp"
#include <stdlib.h>
#include <algorithm>
#include <string.h>
#include <string>
#include <vector>

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <class T>
  T* take(size_t n) {
    off = al256(off);
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct Dims {
  int B, N, R, Np, D, H, I, F, Fp, Th, L, ks, J, E, Ke, Din;  // Din = data width (dim_in), Ke = to_embed input width 2*Din + E
  long M, M0;
};
Dims dims_of(const vbx_model* m) {
  Dims d;
  d.B = m->B; d.N = m->N; d.R = m->R; d.Np = m->N + m->R; d.D = m->D; d.H = m->H; d.I = m->H * 64;
  d.Din = m->Din > 0 ? m->Din : m->D;
  d.E = m->E; d.Ke = 2 * d.Din + m->E;
  d.F = m->F; d.Fp = ((m->F + 63) / 64) * 64; d.Th = m->Th; d.L = m->L; d.ks = m->ksize; d.J = m->L * 4 * m->D;
  d.M = (long)d.B * d.Np; d.M0 = (long)d.B * d.N;
  return d;
}

// *h = fp16 copy (forward NT GEMMs), plain = bf16 copy (backward dgrad NN GEMMs)
struct WLayer {
  u16 *qkv, *qkvh, *out, *outh, *w1, *w1h, *w2, *w2h, *glw, *glwh;
  u16 *skw = nullptr, *skwh = nullptr;  // u-net skip combiner [D, 2D] of the second-half layers
  float* b1;
};
struct WPack {
  u16 *embh, *embb, *pred, *predh, *adah;  // embb: bf16 copy of to_embed (dgrad into cond_emb, text-conditioned models only)
  float* bada;
  std::vector<WLayer> layer;
  size_t bytes;
};
void carve_wpack(const vbx_model* m, WPack& w) {
  const Dims d = dims_of(m);
  Carver c(m->wpack);
  w.embh = c.take<u16>((size_t)d.D * d.Ke);
  w.embb = d.E ? c.take<u16>((size_t)d.D * d.Ke) : nullptr;
  w.pred = c.take<u16>((size_t)d.Din * d.D);
  w.predh = c.take<u16>((size_t)d.Din * d.D);
  w.adah = c.take<u16>((size_t)d.J * d.Th);
  w.bada = c.take<float>(d.J);
  w.layer.resize(d.L);
  for (int l = 0; l < d.L; l++) {
    w.layer[l].qkv = c.take<u16>((size_t)3 * d.I * d.D);
    w.layer[l].qkvh = c.take<u16>((size_t)3 * d.I * d.D);
    w.layer[l].out = c.take<u16>((size_t)d.D * d.I);
    w.layer[l].outh = c.take<u16>((size_t)d.D * d.I);
    w.layer[l].w1 = c.take<u16>((size_t)2 * d.Fp * d.D);
    w.layer[l].w1h = c.take<u16>((size_t)2 * d.Fp * d.D);
    w.layer[l].b1 = c.take<float>(2 * d.Fp);
    w.layer[l].w2 = c.take<u16>((size_t)d.D * d.Fp);
    w.layer[l].w2h = c.take<u16>((size_t)d.D * d.Fp);
    w.layer[l].glw = m->gateloop ? c.take<u16>((size_t)3 * d.D * d.D) : nullptr;
    w.layer[l].glwh = m->gateloop ? c.take<u16>((size_t)3 * d.D * d.D) : nullptr;
    if (m->unet && l >= d.L / 2) {
      w.layer[l].skw = c.take<u16>((size_t)2 * d.D * d.D);
      w.layer[l].skwh = c.take<u16>((size_t)2 * d.D * d.D);
    }
  }
  w.bytes = al256(c.off);
}

struct ALayer {
  u16 *hn1, *hn1h, *q16, *k16, *qb, *kb, *v, *vh, *o, *oh, *hn2, *hn2h, *h1, *g, *gh;
  float *qrn, *krn, *lse;
  unsigned *dbr = nullptr, *dbc = nullptr;  // attention dropout keep bits, row- / column-major (training with attn_dropout > 0)
  // GateLoop: normed input (bf16 | fp16), projection q|kv|a, scan state h, scan output s
  u16 *hg = nullptr, *hgh = nullptr;
  float *glp = nullptr, *glh = nullptr, *gls = nullptr;
};
struct Acts {
  u16 *embed_in, *embed_inh;
  float *e, *four, *pre, *temb, *ada;
  std::vector<float*> xs;  // residual snapshots
  // u-net skip connections (vbx_model.unet): xc[l] = combined input of layer l >= L/2, cat16 / catb = the combiner's [M, 2D] operand
  // (fp16 forward, bf16 recomputed for its weight gradient), dcat = d(cat) fp32, dskip[p] = gradient that reaches the input of
  // layer p < L/2 through its skip, added when the backward gets there
  std::vector<float*> xc, dskip;
  u16 *cat16 = nullptr, *catb = nullptr;
  float* dcat = nullptr;
  std::vector<ALayer> layer;
  u16 *hf, *hfh;
  float *pred, *per_b;
  // backward scratch
  float *dx, *dq, *dk, *delta, *slabs, *npart, *npart2, *cpart, *dada, *dtemb, *cs_scratch, *cs_layers, *gpart, *tmp2d, *ada_scratch, *dpre, *de, *wpart,
      *tscratch;
  u16 *dxb, *dg, *dh1, *dhn, *dO, *dqkv, *deb, *dpb, *gl_dp, *demb;
  char* attn_scratch = nullptr;  // one-pass attention backward: chain flags + running dq sums (vbx_attn_bwd_scratch_bytes)
  u16* dxb2 = nullptr;  // bf16 dx of the attention half, so that FeedForward-out's dx operand survives to the layer's grouped wgrad launch
  float* gl_ds;
  size_t slab_floats;
  size_t np_stride = 0, cp_stride = 0, cs_stride = 0, gp_stride = 0;  // floats per layer region of npart / npart2, cpart, cs_layers, gpart
  size_t bytes;
};

// Split-K factor of a weight-gradient GEMM dW[I,J] (reduction over K = B*Np rows).  Cost model in microseconds, fitted to the
// profiled launches: with W = tiles*s workgroups the busiest CU gets n = ceil(W/256) of them, which (n <= 3) run concurrently and
// share the CU's L2->LDS stream -- a lone workgroup reaches ~60 % of the rate three reach together, two ~85 %; each k-step of 32
// costs ~0.27 us per resident workgroup at full rate; the fp32 slabs cost a write + a read of s*I*J*4 bytes at ~4 TB/s.
// VBX_WGRAD_TARGET=<workgroups> restores the plain "about that many workgroups" rule (A/B).
// The four weight-gradient GEMMs of a layer run as ONE grouped launch at the end of the layer's backward
// (vbx_gemm_tn_splitk_grouped; by default on the 256 x 256 tile of gemm3.hip) instead of four launches interleaved with the dgrads:
// separately they are 8-24 tiles each.  VBX_GROUP_WGRAD=0 restores the separate launches (A/B); with VBX_GEMM3=0 the default is off.
bool group_wgrad() {
  static const char* e = getenv("VBX_GROUP_WGRAD");
  if (e) return atoi(e) == 1;
  return vbx_gemm_path() != 1;
}
struct WgradGroup {
  vbx_gemm_desc d[4];
  int n = 0;
};

// K splits of the grouped launch on 256 x 256 tiles: every job gets the same count, chosen so that the layer's tiles x splits
// fill the 256 CUs once (dim 512: 24 + 8 + 22 + 12 = 66 tiles -> 3 splits = 198 workgroups of ~44 k-tiles).
int wgrad_splits3(const Dims& d) {
  static const long fixed = getenv("VBX_WGRAD_SPLITS3") ? atol(getenv("VBX_WGRAD_SPLITS3")) : 0;
  auto t = [](long I, long J) { return ((I + 255) / 256) * ((J + 255) / 256); };
  const long tiles = t(3 * d.I, d.D) + t(d.D, d.I) + t(2 * d.Fp, d.D) + t(d.D, d.Fp);
  long s = fixed > 0 ? fixed : 256 / tiles;
  const long smax = (d.M + 1023) / 1024;  // at least 16 k-tiles per workgroup
  if (s > smax) s = smax;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return (int)s;
}

int wgrad_splits(long I, long J, long K) {
  const long tiles = ((I + 127) / 128) * ((J + 127) / 128);
  const long smax = (K + 511) / 512;
  static const long target = getenv("VBX_WGRAD_TARGET") ? atol(getenv("VBX_WGRAD_TARGET")) : 0;
  static const long fixed = getenv("VBX_WGRAD_SPLITS") ? atol(getenv("VBX_WGRAD_SPLITS")) : 0;  // A/B: the same split count for every
  if (fixed > 0) return (int)(fixed > smax ? smax : (fixed > 16 ? 16 : fixed));                  // weight gradient (with VBX_GROUP_WGRAD=1
                                                                                                 // occupancy no longer needs many splits)
  if (target > 0) {
    long s = (target + tiles - 1) / tiles;
    if (s > smax) s = smax;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    return (int)s;
  }
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= 16 && s <= smax; s++) {
    const long W = tiles * s;
    const long n = (W + 255) / 256;
    const double eff = n >= 3 ? 1.0 : (n == 2 ? 0.85 : 0.6);
    const double ksteps = (double)((K + s - 1) / s) / 32.0;
    const double gemm = (double)n * ksteps * 0.27 / eff;
    const double slab = (double)s * (double)I * (double)J * 8.0 / 4.0e6;
    const double cost = gemm + slab;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

void carve_acts(const vbx_model* m, Acts& a) {
  const Dims d = dims_of(m);
  Carver c(m->act);
  const bool tr = m->training != 0;
  a.embed_in = tr ? c.take<u16>((size_t)d.M0 * d.Ke) : nullptr;
  a.embed_inh = c.take<u16>((size_t)d.M0 * d.Ke);
  a.e = c.take<float>((size_t)d.M0 * d.D);
  a.four = c.take<float>((size_t)d.B * d.D);
  a.pre = c.take<float>((size_t)d.B * d.Th);
  a.temb = c.take<float>((size_t)d.B * d.Th);
  a.ada = c.take<float>((size_t)d.B * d.J);
  const int S = m->gateloop ? 3 : 2;  // residual updates per layer
  const bool keep = tr || m->unet;  // u-net: the first-half layer inputs are read again by the second half, also in inference
  const int nxs = keep ? S * d.L + 1 : 2;
  a.xs.resize(S * d.L + 1);
  std::vector<float*> bufs(nxs);
  for (int i = 0; i < nxs; i++) bufs[i] = c.take<float>((size_t)d.M * d.D);
  for (int i = 0; i <= S * d.L; i++) a.xs[i] = bufs[keep ? i : (i & 1)];
  a.xc.assign(d.L, nullptr);
  a.dskip.assign(d.L, nullptr);
  if (m->unet) {
    float* shared_xc = nullptr;
    for (int l = d.L / 2; l < d.L; l++) {
      if (tr || !shared_xc) shared_xc = c.take<float>((size_t)d.M * d.D);
      a.xc[l] = shared_xc;
    }
    a.cat16 = c.take<u16>((size_t)d.M * 2 * d.D);
  }
  a.layer.resize(d.L);
  const size_t hs = (size_t)d.B * d.H * d.Np * 64;
  ALayer shared{};
  for (int l = 0; l < d.L; l++) {
    if (tr || l == 0) {
      ALayer& y = a.layer[l];
      y.hn1 = tr ? c.take<u16>((size_t)d.M * d.D) : nullptr;  // bf16 copy: wgrad operand
      y.hn1h = c.take<u16>((size_t)d.M * d.D);                  // fp16: A operand of the q/k/v projection
      y.q16 = c.take<u16>(hs);
      y.k16 = c.take<u16>(hs);
      y.qb = tr ? c.take<u16>(hs) : nullptr;
      y.kb = tr ? c.take<u16>(hs) : nullptr;
      y.v = tr ? c.take<u16>(hs) : nullptr;
      y.vh = c.take<u16>(hs);
      y.qrn = tr ? c.take<float>((size_t)d.B * d.H * d.Np) : nullptr;
      y.krn = tr ? c.take<float>((size_t)d.B * d.H * d.Np) : nullptr;
      y.o = tr ? c.take<u16>((size_t)d.M * d.I) : nullptr;
      y.oh = c.take<u16>((size_t)d.M * d.I);
      y.lse = c.take<float>((size_t)d.B * d.H * d.Np);
      y.hn2 = tr ? c.take<u16>((size_t)d.M * d.D) : nullptr;
      y.hn2h = c.take<u16>((size_t)d.M * d.D);
      y.h1 = tr ? c.take<u16>((size_t)d.M * 2 * d.Fp) : nullptr;
      y.g = tr ? c.take<u16>((size_t)d.M * d.Fp) : nullptr;
      y.gh = c.take<u16>((size_t)d.M * d.Fp);
      if (m->attn_dropout > 0.f) {
        const size_t words = (size_t)d.B * d.H * d.Np * vbx_dropout_bits_words(d.Np);
        y.dbr = c.take<unsigned>(words);
        y.dbc = c.take<unsigned>(words);
      }
      if (m->gateloop) {
        y.hg = tr ? c.take<u16>((size_t)d.M * d.D) : nullptr;
        y.hgh = c.take<u16>((size_t)d.M * d.D);
        y.glp = c.take<float>((size_t)d.M * 3 * d.D);
        y.glh = tr ? c.take<float>((size_t)d.M * d.D) : nullptr;
        y.gls = c.take<float>((size_t)d.M * d.D);
      }
      shared = y;
    } else {
      a.layer[l] = shared;
    }
  }
  a.hf = tr ? c.take<u16>((size_t)d.M0 * d.D) : nullptr;
  a.hfh = c.take<u16>((size_t)d.M0 * d.D);
  a.pred = c.take<float>((size_t)d.M0 * d.Din);
  a.per_b = c.take<float>(vbx_masked_mse_scratch_floats(d.B));
  if (tr) {
    a.dx = c.take<float>((size_t)d.M * d.D);
    a.dxb = c.take<u16>((size_t)d.M * d.D);
    a.dg = c.take<u16>((size_t)d.M * d.Fp);
    a.dh1 = c.take<u16>((size_t)d.M * 2 * d.Fp);
    a.dhn = c.take<u16>((size_t)d.M * d.D);
    a.dO = c.take<u16>((size_t)d.M * d.I);
    a.dqkv = c.take<u16>((size_t)d.M * 3 * d.I);
    a.dq = c.take<float>(hs);
    a.dk = c.take<float>(hs);
    a.delta = c.take<float>((size_t)d.B * d.H * d.Np);
    size_t sf = 0;
    const int s3 = wgrad_splits3(d);
    auto upd = [&](long I, long J, long K) {  // either split scheme may be selected at run time (vbx_gemm_select): size for both
      const int s1 = wgrad_splits(I, J, K);
      const size_t n = (size_t)(s1 > s3 ? s1 : s3) * I * J;
      if (n > sf) sf = n;
    };
    upd(3 * d.I, d.D, d.M); upd(d.D, d.I, d.M); upd(2 * d.Fp, d.D, d.M); upd(d.D, d.Fp, d.M);
    upd(d.D, d.Ke, d.M0); upd(d.Din, d.D, d.M0);
    if (m->gateloop) upd(3 * d.D, d.D, d.M);
    if (m->unet) upd(d.D, 2 * d.D, d.M);
    a.slab_floats = sf;
    a.slabs = c.take<float>(4 * sf);  // four regions: the layer's weight-gradient slabs stay live until its batched reduce
    // partial records of the layer's small gradients: L regions each (vbx_model.defer_reduce: every layer keeps its own until layer 0
    // reduces them all; otherwise region 0 is reused).  The arena layout must not depend on a per-call switch, so the count follows the
    // MODEL only: GateLoop and u-net models never defer (vbx_model_backward_layer) and get one region (at dim 1024 /
    // depth 24 the L regions are several hundred MB).
    const size_t nreg = (m->gateloop || m->unet) ? 1 : (size_t)d.L;
    a.np_stride = (size_t)d.B * vbx_rmsnorm_bwd_chunks(d.Np) * 2 * d.D;  // >= the LayerNorm backward's 16-row records
    a.cp_stride = (size_t)d.B * vbx_rmsnorm_bwd_chunks(d.Np) * d.D;
    a.npart = c.take<float>(a.np_stride * nreg);
    a.npart2 = c.take<float>(a.np_stride * nreg);  // attention pre-norm partials (batched reduce)
    a.cpart = c.take<float>(a.cp_stride * nreg);
    a.dada = c.take<float>((size_t)d.B * d.J);
    a.dtemb = c.take<float>((size_t)d.B * d.Th);
    size_t cs = (size_t)vbx_colsum_scratch_floats((int)d.M, 2 * d.Fp);
    if (cs < (size_t)vbx_geglu_bwd_colsum_slabs() * 2 * d.Fp) cs = (size_t)vbx_geglu_bwd_colsum_slabs() * 2 * d.Fp;
    a.cs_scratch = c.take<float>(cs);
    a.cs_stride = (size_t)vbx_geglu_bwd_colsum_slabs() * 2 * d.Fp;
    a.cs_layers = c.take<float>(a.cs_stride * nreg);
    {
      const int r1 = vbx_qknorm_rope_bwd_gpart_rows(d.B), r2 = d.B * vbx_attn_bwd_fused_tiles(d.Np);
      a.gp_stride = (size_t)2 * (r1 > r2 ? r1 : r2) * d.H * 64;
      a.gpart = c.take<float>(a.gp_stride * nreg);
    }
    a.tmp2d = c.take<float>(2 * d.D);
    a.ada_scratch = c.take<float>(std::max((size_t)vbx_adaln_proj_bwd_scratch_floats(d.B, d.Th, 4 * d.D),
                                           (size_t)vbx_adaln_dtemb_all_scratch_floats(d.L, d.B, d.Th, 4 * d.D)));
    a.dpre = c.take<float>((size_t)d.M0 * d.D);
    a.de = c.take<float>((size_t)d.M0 * d.D);
    a.deb = c.take<u16>((size_t)d.M0 * d.D);
    a.dpb = c.take<u16>((size_t)d.M0 * d.Din);
    a.wpart = c.take<float>((size_t)vbx_convpos_bwd_chunks(d.B, d.N) * d.D * 64);
    a.tscratch = c.take<float>((size_t)vbx_time_embed_bwd_scratch_floats(d.B, d.D));
    a.gl_ds = m->gateloop ? c.take<float>((size_t)d.M * d.D) : nullptr;
    a.gl_dp = m->gateloop ? c.take<u16>((size_t)d.M * 3 * d.D) : nullptr;
    a.demb = d.E ? c.take<u16>((size_t)d.M0 * d.E) : nullptr;
    a.dxb2 = c.take<u16>((size_t)d.M * d.D);  // always carved: the arena layout must not depend on run-time tuning knobs
    a.attn_scratch = c.take<char>(vbx_attn_bwd_scratch_bytes(d.B, d.H, d.Np));
    if (m->unet) {
      a.catb = c.take<u16>((size_t)d.M * 2 * d.D);
      a.dcat = c.take<float>((size_t)d.M * 2 * d.D);
      for (int p = 0; p < d.L / 2; p++) a.dskip[p] = c.take<float>((size_t)d.M * d.D);
    }
  }
  a.bytes = al256(c.off);
}

// ---- in-situ stage timing (vbx_prof_enable / vbx_prof_collect, include/vbx.h)
struct ProfRec {
  const char* label;
  hipEvent_t e0, e1;
};
struct Prof {
  bool on = false;
  std::vector<ProfRec> recs;
};
Prof g_prof;
struct ProfScope {
  hipStream_t st;
  int idx = -1;
  ProfScope(const char* label, hipStream_t st_) : st(st_) {
    if (!g_prof.on) return;
    // never grow without bound (vbx_prof_collect never called) and never record into a capturing stream (an event record would
    // become a graph node and the elapsed-time query on it fails): profiling simply pauses there
    if (g_prof.recs.size() >= 8192) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    ProfRec r{label, nullptr, nullptr};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    idx = (int)g_prof.recs.size();
    g_prof.recs.push_back(r);
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(g_prof.recs[idx].e1, st);
  }
};

#define CK(x)                 \
  do {                        \
    int rc__ = (x);           \
    if (rc__ != 0) return rc__; \
  } while (0)

int check_model(const vbx_model* m) {
  VBX_REQUIRE(m && m->params && m->off && m->wpack && m->act && m->rot_cos && m->rot_sin, "vbx_model: null field");
  VBX_REQUIRE(m->D % 64 == 0 && m->D <= 2048, "vbx_model: dim must be a multiple of 64 and <= 2048 (got %d)", m->D);
  VBX_REQUIRE(m->H > 0 && m->H % 2 == 0, "vbx_model: heads must be even (dim_head is fixed at 64)");
  VBX_REQUIRE(m->Th % 8 == 0 && m->L > 0 && m->B > 0 && m->N > 0 && m->R >= 0 && m->F > 0, "vbx_model: bad dims");
  VBX_REQUIRE(m->ksize >= 1 && m->ksize <= 31 && (m->ksize & 1), "vbx_model: conv_pos_embed_kernel_size must be odd and <= 31 (got %d)", m->ksize);
  VBX_REQUIRE(m->E >= 0 && m->E % 8 == 0 && (m->E == 0 || (m->V1 > 0 && !m->stack_only)), "vbx_model: bad dim_cond_emb / table size");
  const long DT = (long)m->D * m->Th;
  for (int l = 0; l < m->L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    if (m->plain_norm) continue;
    VBX_REQUIRE(o[VBX_L_B1W] == o[VBX_L_G1W] + DT && o[VBX_L_G2W] == o[VBX_L_B1W] + DT && o[VBX_L_B2W] == o[VBX_L_G2W] + DT,
                "vbx_model: adaLN weights of layer %d are not contiguous in (g1,b1,g2,b2) order", l);
    VBX_REQUIRE(o[VBX_L_B1B] == o[VBX_L_G1B] + m->D && o[VBX_L_G2B] == o[VBX_L_B1B] + m->D && o[VBX_L_B2B] == o[VBX_L_G2B] + m->D,
                "vbx_model: adaLN biases of layer %d are not contiguous in (g1,b1,g2,b2) order", l);
  }
  for (int l = 0; l < m->L && m->gateloop; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    VBX_REQUIRE(o[VBX_L_GLLNB] == o[VBX_L_GLLNW] + m->D,
                "vbx_model: GateLoop post-LayerNorm weight and bias of layer %d are not contiguous", l);
  }
  VBX_REQUIRE(!m->plain_norm || m->stack_only, "vbx_model: plain_norm is only used by the standalone stack (VoiceBox is adaptive)");
  VBX_REQUIRE(m->Din >= 0 && m->Din % 8 == 0 && (m->Din == 0 || !m->stack_only), "vbx_model: dim_in must be a multiple of 8 (got %d)", m->Din);
  VBX_REQUIRE(m->attn_dropout >= 0.f && m->attn_dropout < 1.f && m->ff_dropout >= 0.f && m->ff_dropout < 1.f, "vbx_model: dropout must be in [0, 1)");
  VBX_REQUIRE(!m->unet || (m->stack_only && m->L % 2 == 0 && !m->precise),
              "vbx_model: u-net skip connections belong to the standalone stack (even depth); VoiceBox never enables them");
  return 0;
}

// input of layer l as its blocks see it: the u-net combiner's output in the second half, else the previous layer's output
float* layer_input(const vbx_model* m, const Acts& a, int l) {
  const int S = m->gateloop ? 3 : 2;
  return (m->unet && l >= m->L / 2) ? a.xc[l] : a.xs[S * l];
}

// forward GEMMs: fp16 operands
int gemm_nt(const u16* A, int lda, const u16* Bw, int ldb, int M, int N, int K, int epi, void* C, int ldc, const float* bias,
            const float* resid, void* C2, void* C3, hipStream_t st) {
  vbx_gemm_desc g{};
  g.mode = VBX_GEMM_NT; g.epilogue = epi; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.A = A; g.B = Bw; g.C = C; g.bias = bias; g.resid = resid; g.C2 = C2; g.C3 = C3; g.f16 = 1;
  return vbx_gemm(&g, st);
}
int gemm_nn_bf16(const u16* A, int lda, const u16* Bw, int ldb, int M, int N, int K, u16* C, int ldc, hipStream_t st) {
  vbx_gemm_desc g{};
  g.mode = VBX_GEMM_NN; g.epilogue = VBX_EPI_BF16; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.A = A; g.B = Bw; g.C = C;
  return vbx_gemm(&g, st);
}
// Weight gradients are leaves of the backward graph: nothing downstream in the layer needs them, so they CAN run on a side stream
// (forked from / joined to the caller's stream with events) to fill the CUs the dependent chain leaves idle.  Opt-in
// (VBX_WGRAD_STREAM=1): measured in the same run the train step got 4 % SLOWER (14.40 -> 14.97 ms) -- the split-K GEMMs and the
// dgrad / attention kernels they overlap with are all bound by the same L2->LDS stream, so concurrency only adds contention.
// One side stream per process, in-order, so the shared split-K slab buffer needs no extra protection.  wgrad_join() makes the
// caller's stream wait for everything issued so far: called before a kernel overwrites an operand a pending wgrad reads
// (a.dxb) and at the end of every stage entry point (the caller may all-reduce / apply the gradients right after it returns).
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, done = nullptr;
  bool pending = false, ok = false;
};
SideStream& side_stream() {
  static thread_local SideStream ss;
  static const bool enabled = getenv("VBX_WGRAD_STREAM") && atoi(getenv("VBX_WGRAD_STREAM")) != 0;
  if (enabled && !ss.s) {
    ss.ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.done, hipEventDisableTiming) == hipSuccess;
  }
  return ss;
}
// Forward: the time embedding + the adaLN projections of the whole stack (a 100 MB weight stream, ~40 us, HBM-bound) do not depend
// on the frame embedding (pack + to_embed GEMM + conv), so they CAN run on their own stream between a fork and a join event --
// also under hipGraph capture, where the two become parallel branches.  Opt-in (VBX_TIME_BRANCH=1): measured in the same run
// the 128-forward sample got 1 % slower (365 -> 369 ms) and the train step 1.6 % slower (12.90 -> 13.12 ms).
struct TimeBranch {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, done = nullptr;
  bool ok = false;
};
TimeBranch& time_branch() {
  static thread_local TimeBranch tb;
  static const bool enabled = getenv("VBX_TIME_BRANCH") && atoi(getenv("VBX_TIME_BRANCH")) != 0;
  if (enabled && !tb.s) {
    tb.ok = hipStreamCreateWithFlags(&tb.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&tb.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&tb.done, hipEventDisableTiming) == hipSuccess;
  }
  return tb;
}
int wgrad_join(hipStream_t st) {
  SideStream& ss = side_stream();
  if (!ss.ok || !ss.pending) return 0;
  if (hipStreamWaitEvent(st, ss.done, 0) != hipSuccess) {
    vbx_set_error("wgrad_join: hipStreamWaitEvent failed");
    return VBX_EINVAL;
  }
  ss.pending = false;
  return 0;
}
// dW[I,J] = P[K,I]^T . Q[K,J]  -> grads (fp32, reference layout [dst_rows, dst_cols])
int wgrad(const u16* P, int ldp, const u16* Q, int ldq, int I, int J, long K, float* slabs, float* dst, int dst_rows,
          int dst_cols, int rowmap, int F, hipStream_t st, vbx_skr_jobs* defer = nullptr, WgradGroup* grp = nullptr, int group_splits = 0,
          float* sq = nullptr /* vbx_skr_job.sq of the deferred reduction */) {
  vbx_gemm_desc g{};
  const bool grouped = grp && defer && J % 4 == 0 && defer->n < VBX_SKR_MAX && grp->n < 4;
  const int splits = (grouped && group_splits > 0) ? group_splits : wgrad_splits(I, J, K);
  g.mode = VBX_GEMM_TN; g.epilogue = VBX_EPI_SPLITK; g.M = I; g.N = J; g.K = (int)K; g.lda = ldp; g.ldb = ldq;
  g.A = P; g.B = Q; g.C = slabs; g.splits = splits;
  if (grouped) {  // GEMM and reduction both deferred to the layer's end
    grp->d[grp->n++] = g;
    vbx_skr_job& jb = defer->job[defer->n++];
    jb.slabs = slabs; jb.dst = dst; jb.splits = splits; jb.M = I; jb.N = J; jb.dst_rows = dst_rows; jb.dst_cols = dst_cols;
    jb.dst_ld = dst_cols; jb.rowmap = rowmap; jb.F = F; jb.sq = sq;
    return 0;
  }
  SideStream& ss = side_stream();
  hipStream_t run = st;
  if (ss.ok) {
    if (hipEventRecord(ss.fork, st) != hipSuccess || hipStreamWaitEvent(ss.s, ss.fork, 0) != hipSuccess) {
      vbx_set_error("wgrad: fork to the side stream failed");
      return VBX_EINVAL;
    }
    run = ss.s;
  }
  { ProfScope ps("wgrad (separate)", run); CK(vbx_gemm(&g, run)); }
  if (defer && J % 4 == 0 && defer->n < VBX_SKR_MAX) {  // reduced later, together with the layer's other weight gradients
    vbx_skr_job& jb = defer->job[defer->n++];
    jb.slabs = slabs; jb.dst = dst; jb.splits = splits; jb.M = I; jb.N = J; jb.dst_rows = dst_rows; jb.dst_cols = dst_cols;
    jb.dst_ld = dst_cols; jb.rowmap = rowmap; jb.F = F; jb.sq = sq;
  } else {
    CK(vbx_splitk_reduce(slabs, splits, I, J, dst, dst_rows, dst_cols, dst_cols, rowmap, F, 0, run));
  }
  if (ss.ok) {
    if (hipEventRecord(ss.done, ss.s) != hipSuccess) {
      vbx_set_error("wgrad: event record on the side stream failed");
      return VBX_EINVAL;
    }
    ss.pending = true;
  }
  return 0;
}

}  // namespace

extern "C" size_t vbx_model_wpack_bytes(const vbx_model* m) {
  vbx_model t = *m;
  t.wpack = nullptr;
  WPack w;
  carve_wpack(&t, w);
  return w.bytes;
}
extern "C" size_t vbx_model_act_bytes(const vbx_model* m) {
  vbx_model t = *m;
  t.act = nullptr;
  Acts a;
  carve_acts(&t, a);
  return a.bytes;
}

extern "C" int vbx_model_pack_weights(const vbx_model* m, void* stream) {
  CK(check_model(m));
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  const float* P = m->params;
  const long* G = m->off;
  if (!m->stack_only) {
    CK(vbx_pack_weight(P + G[VBX_P_EMBW], d.D, d.Ke, w.embb, w.embh, d.D, d.Ke, 0, 0, stream));
    CK(vbx_pack_weight(P + G[VBX_P_PREDW], d.Din, d.D, w.pred, w.predh, d.Din, d.D, 0, 0, stream));
  }
  for (int l = 0; l < d.L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    if (!m->plain_norm) {
      CK(vbx_pack_weight(P + o[VBX_L_G1W], 4 * d.D, d.Th, nullptr, w.adah + (size_t)l * 4 * d.D * d.Th, 4 * d.D, d.Th, 0, 0, stream));
      CK(vbx_pack_bias(P + o[VBX_L_G1B], 4 * d.D, w.bada + (size_t)l * 4 * d.D, 4 * d.D, 0, 0, stream));
    }
    CK(vbx_pack_weight(P + o[VBX_L_QKVW], 3 * d.I, d.D, w.layer[l].qkv, w.layer[l].qkvh, 3 * d.I, d.D, 0, 0, stream));
    CK(vbx_pack_weight(P + o[VBX_L_OUTW], d.D, d.I, w.layer[l].out, w.layer[l].outh, d.D, d.I, 0, 0, stream));
    CK(vbx_pack_weight(P + o[VBX_L_FF1W], 2 * d.F, d.D, w.layer[l].w1, w.layer[l].w1h, 2 * d.Fp, d.D, 1, d.F, stream));
    CK(vbx_pack_bias(P + o[VBX_L_FF1B], 2 * d.F, w.layer[l].b1, 2 * d.Fp, 1, d.F, stream));
    CK(vbx_pack_weight(P + o[VBX_L_FF2W], d.D, d.F, w.layer[l].w2, w.layer[l].w2h, d.D, d.Fp, 0, 0, stream));
    if (m->gateloop) CK(vbx_pack_weight(P + o[VBX_L_GLW], 3 * d.D, d.D, w.layer[l].glw, w.layer[l].glwh, 3 * d.D, d.D, 0, 0, stream));
    if (w.layer[l].skw) CK(vbx_pack_weight(P + o[VBX_L_SKW], d.D, 2 * d.D, w.layer[l].skw, w.layer[l].skwh, d.D, 2 * d.D, 0, 0, stream));
  }
  return 0;
}

// adaLN projections of n <= 16 conditioning rows temb [n, Th] with the model's packed weights -> ada [L][n][4 * D]: what
// vbx_model_forward computes per forward, exposed so that the sampler can tabulate it over its whole time grid (vbx_ada_select).
extern "C" int vbx_model_adaln_table(const vbx_model* m, const float* temb, int n, float* ada, void* stream) {
  CK(check_model(m));
  VBX_REQUIRE(temb && ada && n > 0 && n <= 16 && !m->plain_norm, "vbx_model_adaln_table: bad args");
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  if (m->precise) {
    const float* P = m->params;
    for (int l = 0; l < d.L; l++) {
      const long* o = m->off + VBX_NG + (long)l * VBX_NL;
      CK(vbx_adaln_proj_f32(temb, P + o[VBX_L_G1W], P + o[VBX_L_G1B], ada + (size_t)l * n * 4 * d.D, n, d.Th, 4 * d.D, 4 * d.D, stream));
    }
    return 0;
  }
  return vbx_adaln_proj_fwd(temb, w.adah, w.bada, ada, n, d.Th, d.J, 4 * d.D, stream);
}

extern "C" int vbx_model_forward(const vbx_model* m, const vbx_io* io, void* stream) {
  CK(check_model(m));
  VBX_REQUIRE(io && io->x, "vbx_model_forward: null io field");
  VBX_REQUIRE(m->stack_only ? ((m->plain_norm || io->cond) && io->pred && !io->target) : (io->cond && (io->times || io->ada_table)),
              "vbx_model_forward: null io field");
  VBX_REQUIRE(!io->target || (io->loss_mask && io->loss), "vbx_model_forward: target needs loss_mask and loss");
  VBX_REQUIRE((io->attn_mask == nullptr) == (io->attn_mask_p == nullptr), "vbx_model_forward: attn_mask and attn_mask_p go together");
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  Acts a;
  carve_acts(m, a);
  const float* P = m->params;
  const long* G = m->off;
  const bool tr = m->training != 0;

  if (m->precise) {  // exact-operand forward (precise.hip) into the same arenas: the backward entry points read them as usual
    std::vector<VbxPreciseLayer> pl(d.L);
    for (int l = 0; l < d.L; l++) {
      const ALayer& y = a.layer[l];
      pl[l] = VbxPreciseLayer{y.hn1, y.q16, y.k16, y.qb, y.kb, y.v, y.vh, y.oh, y.o, y.hn2, y.gh, y.g, tr ? y.h1 : nullptr,
                              y.qrn, y.krn, y.lse, w.layer[l].b1, y.hg, y.glp, y.gls, tr ? y.glh : nullptr, (unsigned*)y.dbr, (unsigned*)y.dbc};
    }
    VbxPreciseActs pa{a.four, a.pre, a.temb, a.ada, a.e, a.pred, a.per_b, a.xs.data(), pl.data(), a.embed_in, a.embed_inh, a.hf};
    return vbx_forward_precise(m, io, &pa, stream);
  }

  if (m->stack_only) {
    // standalone Transformer.forward: registers + x, the caller's condition drives the adaLN projections  (:417-431, :449-451)
    CK(vbx_stack_input(io->x, d.R ? P + G[VBX_P_REG] : nullptr, a.xs[0], d.B, d.N, d.R, d.D, stream));
    if (!m->plain_norm) CK(vbx_adaln_proj_fwd(io->cond, w.adah, w.bada, a.ada, d.B, d.Th, d.J, 4 * d.D, stream));
  } else {
  // time embedding + every adaLN projection of the stack   (:1082, :273) -- on the side branch when available
  TimeBranch& tb = time_branch();
  void* tstream = stream;
  if (tb.ok) {
    if (hipEventRecord(tb.fork, st) != hipSuccess || hipStreamWaitEvent(tb.s, tb.fork, 0) != hipSuccess) {
      vbx_set_error("vbx_model_forward: fork of the time branch failed");
      return VBX_EINVAL;
    }
    tstream = tb.s;
  }
  if (io->ada_table) {  // sampler: the projections of this time point were evaluated once for the whole grid (vbx_ada_select)
    VBX_REQUIRE(!tr && io->ada_counter, "vbx_model_forward: ada_table is an inference-only input and needs ada_counter");
    CK(vbx_ada_select(a.ada, d.L, d.B, 4 * d.D, io->ada_table, io->ada_counter, io->ada_slot, tstream));
  } else {
  CK(vbx_time_embed_fwd(io->times, P + G[VBX_P_SINW], P + G[VBX_P_T1W], P + G[VBX_P_T1B], a.four, a.pre, a.temb, d.B, d.D,
                        d.Th, tstream));
  CK(vbx_adaln_proj_fwd(a.temb, w.adah, w.bada, a.ada, d.B, d.Th, d.J, 4 * d.D, tstream));
  }
  if (tb.ok && hipEventRecord(tb.done, tb.s) != hipSuccess) {
    vbx_set_error("vbx_model_forward: event record on the time branch failed");
    return VBX_EINVAL;
  }
  // to_embed(cat(x, cond * ~cond_mask))   (voicebox_pytorch.py:1035,1075-1078)
  if (d.E) {
    VBX_REQUIRE(io->cond_ids && io->T > 0, "vbx_model_forward: a text-conditioned model needs cond_ids");
    CK(vbx_pack_embed_input_text(io->x, io->cond, io->cond_mask, io->drop_mask, io->null_cond, io->cond_ids, io->T,
                                 P + G[VBX_P_CEMB], d.E, io->null_id, a.embed_inh, a.embed_in, d.B, d.N, d.Din, stream));
  } else {
    CK(vbx_pack_embed_input(io->x, io->cond, io->cond_mask, a.embed_inh, a.embed_in, d.B, d.N, d.Din, stream));
  }
  CK(gemm_nt(a.embed_inh, d.Ke, w.embh, d.Ke, (int)d.M0, d.D, d.Ke, VBX_EPI_F32, a.e, d.D, P + G[VBX_P_EMBB], nullptr,
             nullptr, nullptr, st));
  // conv_embed(x) + x, register tokens in place   (:1080, :422-425)
  CK(vbx_convpos_fwd(a.e, P + G[VBX_P_CONVW], P + G[VBX_P_CONVB], io->attn_mask, d.R ? P + G[VBX_P_REG] : nullptr, a.xs[0],
                     d.B, d.N, d.R, d.D, d.ks, stream));
  if (tb.ok && hipStreamWaitEvent(st, tb.done, 0) != hipSuccess) {  // join: the layers read a.ada
    vbx_set_error("vbx_model_forward: join of the time branch failed");
    return VBX_EINVAL;
  }
  }

  for (int l = 0; l < d.L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    const ALayer& y = a.layer[l];
    const float* ada_l = a.ada + (size_t)l * d.B * 4 * d.D;  // [B][g1|b1|g2|b2]
    const int S = m->gateloop ? 3 : 2;
    float* x0 = layer_input(m, a, l);
    float* x_in = m->gateloop ? a.xs[S * l + 1] : x0;
    float* x_mid = a.xs[S * l + S - 1];
    float* x_out = a.xs[S * l + S];
    if (m->unet && l >= d.L / 2) {
      // x = skip_combiner(cat(x, skip * skip_connect_scale))   (:458-463); the skip is the input of layer L-1-l
      CK(vbx_unet_cat(a.xs[S * l], a.xs[S * (d.L - 1 - l)], m->skip_scale, a.cat16, nullptr, d.M, d.D, stream));
      CK(gemm_nt(a.cat16, 2 * d.D, w.layer[l].skwh, 2 * d.D, (int)d.M, d.D, 2 * d.D, VBX_EPI_F32, x0, d.D, P + o[VBX_L_SKB], nullptr,
                 nullptr, nullptr, st));
    }
    if (m->gateloop) {
      // x = GateLoop(x) + x   (:465-466): RMSNorm -> to_qkva -> gated scan -> post LayerNorm + residual
      CK(vbx_rmsnorm_fwd(x0, P + o[VBX_L_GLG], nullptr, 0, y.hg, y.hgh, d.B, d.Np, 0, d.Np, d.D, stream));
      CK(gemm_nt(y.hgh, d.D, w.layer[l].glwh, d.D, (int)d.M, 3 * d.D, d.D, VBX_EPI_F32, y.glp, 3 * d.D, nullptr, nullptr, nullptr,
                 nullptr, st));
      CK(vbx_gateloop_scan_fwd(y.glp, y.gls, tr ? y.glh : nullptr, d.B, d.Np, d.D, stream));
      CK(vbx_layernorm_fwd(y.gls, P + o[VBX_L_GLLNW], P + o[VBX_L_GLLNB], x0, x_in, d.M, d.D, 1e-5f, stream));
    }
    // attn_prenorm -> to_qkv (+qk-norm, rotary) -> Attend -> to_out + residual   (:468-469, :317-333)
    if (m->plain_norm) CK(vbx_rmsnorm_fwd(x_in, P + o[VBX_L_N1G], nullptr, 0, y.hn1, y.hn1h, d.B, d.Np, 0, d.Np, d.D, stream));
    else CK(vbx_rmsnorm_fwd(x_in, ada_l, ada_l + d.D, 4 * d.D, y.hn1, y.hn1h, d.B, d.Np, 0, d.Np, d.D, stream));
    vbx_gemm_desc g{};
    g.mode = VBX_GEMM_NT; g.epilogue = VBX_EPI_QKV; g.M = (int)d.M; g.N = 3 * d.I; g.K = d.D; g.lda = d.D; g.ldb = d.D;
    g.A = y.hn1h; g.B = w.layer[l].qkvh; g.f16 = 1; g.Np = d.Np; g.H = d.H; g.qk_scale = m->qk_norm ? 8.0f : 0.0f;
    g.q_gamma = m->qk_norm ? P + o[VBX_L_QG] : nullptr;
    g.k_gamma = m->qk_norm ? P + o[VBX_L_KG] : nullptr;
    g.rot_cos = m->rot_cos; g.rot_sin = m->rot_sin;
    g.q16 = y.q16; g.k16 = y.k16; g.qb = y.qb; g.kb = y.kb; g.v = y.v; g.v16 = y.vh; g.q_rnorm = y.qrn; g.k_rnorm = y.krn;
    g.q_prescale = vbx_attn_q_prescale(m->attn_scale);  // q16 in the exp2 domain: the attention kernels' contract (include/vbx.h)
    { ProfScope ps("fwd to_qkv", st); CK(vbx_gemm(&g, stream)); }
    const bool drop_on = io->dropout != 0;
    if (y.dbr && drop_on) {  // attend.py:131: keep bits of this layer, kept in the arena for the backward
      CK(vbx_attn_dropout_bits(y.dbr, y.dbc, d.B * d.H, d.Np, io->drop_seed, 2u * l, m->attn_dropout, stream));
      ProfScope ps("fwd attention", st);
      CK(vbx_attn_fwd_dropout(y.q16, y.k16, y.vh, io->attn_mask_p, y.oh, y.o, y.lse, d.B, d.H, d.Np, m->attn_scale, y.dbr,
                              m->attn_dropout, stream));
    } else {
      ProfScope ps("fwd attention", st);
      CK(vbx_attn_fwd(y.q16, y.k16, y.vh, io->attn_mask_p, y.oh, y.o, y.lse, d.B, d.H, d.Np, m->attn_scale, stream));
    }
    { ProfScope ps("fwd to_out", st);
      CK(gemm_nt(y.oh, d.I, w.layer[l].outh, d.I, (int)d.M, d.D, d.I, VBX_EPI_F32, x_mid, d.D, nullptr, x_in, nullptr, nullptr, st)); }
    // ff_prenorm -> FeedForward (GEGLU) + residual   (:471-472, :337-349)
    if (m->plain_norm) CK(vbx_rmsnorm_fwd(x_mid, P + o[VBX_L_N2G], nullptr, 0, y.hn2, y.hn2h, d.B, d.Np, 0, d.Np, d.D, stream));
    else CK(vbx_rmsnorm_fwd(x_mid, ada_l + 2 * d.D, ada_l + 3 * d.D, 4 * d.D, y.hn2, y.hn2h, d.B, d.Np, 0, d.Np, d.D, stream));
    { ProfScope ps("fwd ff_in", st);
      CK(gemm_nt(y.hn2h, d.D, w.layer[l].w1h, d.D, (int)d.M, 2 * d.Fp, d.D, VBX_EPI_GEGLU, y.gh, d.Fp, w.layer[l].b1, nullptr,
                 tr ? y.h1 : nullptr, y.g, st)); }
    if (drop_on && m->ff_dropout > 0.f)  // nn.Dropout between GEGLU and the output projection (:346): both copies of the GEGLU output
      CK(vbx_dropout_rows(y.gh, tr ? y.g : nullptr, d.M, d.Fp, d.Fp, io->drop_seed, 2u * l + 1u, m->ff_dropout, stream));
    { ProfScope ps("fwd ff_out", st);
      CK(gemm_nt(y.gh, d.Fp, w.layer[l].w2h, d.Fp, (int)d.M, d.D, d.Fp, VBX_EPI_F32, x_out, d.D, P + o[VBX_L_FF2B], x_mid, nullptr,
                 nullptr, st)); }
  }
  if (m->stack_only)  // strip registers, final RMSNorm -> fp32 output   (:476-479)
    return vbx_rmsnorm_fwd_f32(a.xs[(m->gateloop ? 3 : 2) * d.L], P + G[VBX_P_FNG], nullptr, 0, io->pred, d.B, d.Np, d.R, d.N, d.D,
                               stream);
  // strip registers, final RMSNorm, to_pred   (:476-479, :1092)
  CK(vbx_rmsnorm_fwd(a.xs[(m->gateloop ? 3 : 2) * d.L], P + G[VBX_P_FNG], nullptr, 0, a.hf, a.hfh, d.B, d.Np, d.R, d.N, d.D, stream));
  float* pred = io->pred ? io->pred : a.pred;
  CK(gemm_nt(a.hfh, d.D, w.predh, d.D, (int)d.M0, d.Din, d.D, VBX_EPI_F32, pred, d.Din, nullptr, nullptr, nullptr, nullptr, st));
  if (io->target) CK(vbx_masked_mse_fwd(pred, io->target, io->loss_mask, a.per_b, io->loss, d.B, d.N, d.Din, stream));
  return 0;
}

static int backward_head_impl(const vbx_model* m, const vbx_io* io, const float* gscale, void* stream) {
  CK(check_model(m));
  VBX_REQUIRE(m->training && m->grads && io && io->target && (m->stack_only || io->loss_mask),
              "vbx_model_backward_head: needs a training forward");
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  Acts a;
  carve_acts(m, a);
  const float* P = m->params;
  float* Gd = m->grads;
  const long* G = m->off;
  if (m->stack_only) {
    // io->target = d(output) fp32 [B,N,D] -> bf16 operand of the final-norm backward
    CK(vbx_pack_weight(io->target, (int)d.M0, d.D, a.dhn, nullptr, (int)d.M0, d.D, 0, 0, stream));
  } else {
    const float* pred = io->pred ? io->pred : a.pred;
    CK(vbx_masked_mse_bwd(pred, io->target, io->loss_mask, a.per_b, gscale, nullptr, a.dpb, d.B, d.N, d.Din, stream));
    CK(wgrad(a.dpb, d.Din, a.hf, d.D, d.Din, d.D, d.M0, a.slabs, Gd + G[VBX_P_PREDW], d.Din, d.D, 0, 0, st));
    CK(gemm_nn_bf16(a.dpb, d.Din, w.pred, d.D, (int)d.M0, d.D, d.Din, a.dhn, d.D, st));
  }
  // gradient wrt the last residual snapshot: zero at the register rows, final-norm backward elsewhere
  if (hipMemsetAsync(a.dx, 0, (size_t)d.M * d.D * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(a.dxb, 0, (size_t)d.M * d.D * sizeof(u16), st) != hipSuccess) {
    vbx_set_error("vbx_model_backward_head: memset failed");
    return VBX_EINVAL;
  }
  CK(vbx_rmsnorm_bwd(a.xs[(m->gateloop ? 3 : 2) * d.L], P + G[VBX_P_FNG], 0, a.dhn, nullptr, a.dx, a.dxb, a.npart, nullptr, d.B, d.Np, d.R, d.N, d.D,
                     stream));
  CK(vbx_reduce_norm_partials(a.npart, a.tmp2d, 0, d.B, vbx_rmsnorm_bwd_chunks(d.N), d.D, 1, stream));
  CK(vbx_sum_rows_f32(a.tmp2d, 1, d.D, Gd + G[VBX_P_FNG], d.D, 0, stream));
  return 0;
}

// the four split-K weight-gradient reductions of a layer are deferred into one launch (own slab region each); VBX_BATCH_WGRAD=0: A/B
static bool batch_wgrad_on() {
  static const bool on = !(getenv("VBX_BATCH_WGRAD") && atoi(getenv("VBX_BATCH_WGRAD")) == 0) && !side_stream().ok;
  return on;
}
// vbx_model.sq_partials: blocks of the layer's four deferred reductions in the order backward_layer_impl issues them
struct SqLayout { long off[4], per_layer; };  // FeedForward-out, FeedForward-in, to_out, to_qkv
static SqLayout sq_layout(const Dims& d) {
  SqLayout q;
  const long n[4] = {vbx_splitk_reduce_blocks(d.D, d.Fp), vbx_splitk_reduce_blocks(2 * d.Fp, d.D), vbx_splitk_reduce_blocks(d.D, d.I),
                     vbx_splitk_reduce_blocks(3 * d.I, d.D)};
  q.off[0] = 0;
  for (int i = 1; i < 4; i++) q.off[i] = q.off[i - 1] + n[i - 1];
  q.per_layer = q.off[3] + n[3];
  return q;
}
extern "C" long vbx_model_sq_partials(const vbx_model* m, long* ranges) {
  if (check_model(m) != 0) return 0;
  if (!batch_wgrad_on()) return 0;
  const Dims d = dims_of(m);
  if (ranges) {
    for (int l = 0; l < d.L; l++) {  // the flat layout holds them in this order (engine.py L_NAMES / VBX_L_*)
      const long* o = m->off + VBX_NG + (long)l * VBX_NL;
      const long lo[4] = {o[VBX_L_QKVW], o[VBX_L_OUTW], o[VBX_L_FF1W], o[VBX_L_FF2W]};
      const long sz[4] = {3L * d.I * d.D, (long)d.D * d.I, 2L * d.F * d.D, (long)d.D * d.F};
      for (int i = 0; i < 4; i++) {
        if (i && lo[i] < lo[i - 1] + sz[i - 1]) return 0;  // an unexpected layout: do not serve it
        ranges[(4 * l + i) * 2] = lo[i];
        ranges[(4 * l + i) * 2 + 1] = lo[i] + sz[i];
      }
    }
  }
  return sq_layout(d).per_layer * d.L;
}

static int backward_layer_impl(const vbx_model* m, const vbx_io* io, int l, void* stream) {
  CK(check_model(m));
  VBX_REQUIRE(m->training && m->grads && l >= 0 && l < m->L, "vbx_model_backward_layer: bad layer / not training");
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  Acts a;
  carve_acts(m, a);
  const float* P = m->params;
  float* Gd = m->grads;
  const long* o = m->off + VBX_NG + (long)l * VBX_NL;
  const ALayer& y = a.layer[l];
  const float* ada_l = a.ada + (size_t)l * d.B * 4 * d.D;
  float* dada_l = a.dada + (size_t)l * d.B * 4 * d.D;
  const int chunks = vbx_rmsnorm_bwd_chunks(d.Np), ln_chunks = (d.Np + 15) / 16;
  const int M = (int)d.M;
  const int S = m->gateloop ? 3 : 2;
  const float* x_in = m->gateloop ? a.xs[S * l + 1] : layer_input(m, a, l);   // input of the attention block
  const float* x_mid = a.xs[S * l + S - 1];  // input of the feed-forward block

  static const bool batched = !(getenv("VBX_BATCH_REDUCE") && atoi(getenv("VBX_BATCH_REDUCE")) == 0);  // 0: one launch per reduction (A/B)
  // adaLN weight gradients in factor form: nothing of the projections' backward is left per layer (bias gradient: two jobs of the
  // batched reduce; d(time_emb): one launch for all layers after layer 0) -- VBX_ADALN_BWD_ALL=0: the per-layer launch (A/B)
  static const bool ada_all_env = !(getenv("VBX_ADALN_BWD_ALL") && atoi(getenv("VBX_ADALN_BWD_ALL")) == 0);
  const bool ada_all = ada_all_env && batched && m->adaln_factors && !m->plain_norm;
  // vbx_model.defer_reduce: this layer's partial records stay in its own region; layer 0 reduces every layer's (VBX_DEFER_REDUCE=0: A/B).
  // GateLoop re-uses npart for its own immediate reductions and the u-net combiner cs_scratch: those models reduce per layer.
  static const bool defer_env = !(getenv("VBX_DEFER_REDUCE") && atoi(getenv("VBX_DEFER_REDUCE")) == 0);
  // The per-layer adaLN backward (materialised weight gradients) consumes dada_l right away: it needs the per-layer reduce too.
  const bool defer = defer_env && batched && m->defer_reduce && !m->gateloop && !m->unet && (m->plain_norm || ada_all);
  const int rr = defer ? l : 0;
  float* const npart = a.npart + rr * a.np_stride;
  float* const npart2 = a.npart2 + rr * a.np_stride;
  float* const cpart = a.cpart + rr * a.cp_stride;
  float* const gpart = a.gpart + rr * a.gp_stride;
  float* const csl = defer ? a.cs_layers + rr * a.cs_stride : a.cs_scratch;  // GEGLU-backward column-sum slabs
  const bool batch_wg = batch_wgrad_on();
  const SqLayout sql = sq_layout(d);
  float* const sqb = (m->sq_partials && batch_wg) ? m->sq_partials + (long)l * sql.per_layer : nullptr;
  auto sq_at = [&](int i) { return sqb ? sqb + sql.off[i] : nullptr; };
  vbx_skr_jobs wj{};
  vbx_skr_jobs* wjp = batch_wg ? &wj : nullptr;
  WgradGroup wgg;
  WgradGroup* wgp = (batch_wg && group_wgrad() && a.dxb2) ? &wgg : nullptr;
  const int gs = (wgp && vbx_gemm_path() != 1) ? wgrad_splits3(d) : 0;  // split count of the grouped 256-tile launch
  u16* dxb_attn = wgp ? a.dxb2 : a.dxb;  // bf16 dx entering the attention half (see Acts::dxb2)
  const size_t sfl = a.slab_floats;
  // ---- FeedForward
  { ProfScope ps("dgrad ff_out", st); CK(gemm_nn_bf16(a.dxb, d.D, w.layer[l].w2, d.Fp, M, d.Fp, d.D, a.dg, d.Fp, st)); }
  CK(wgrad(a.dxb, d.D, y.g, d.Fp, d.D, d.Fp, d.M, a.slabs, Gd + o[VBX_L_FF2W], d.D, d.F, 0, 0, st, wjp, wgp, gs, sq_at(0)));
  VBX_REQUIRE(io || !(m->ff_dropout > 0.f || m->attn_dropout > 0.f), "vbx_model_backward_layer: a model with dropout needs the forward's io");
  const bool drop_on = io && io->dropout != 0;
  if (drop_on && m->ff_dropout > 0.f)  // the same mask on the gradient of the GEGLU output
    CK(vbx_dropout_rows(nullptr, a.dg, d.M, d.Fp, d.Fp, io->drop_seed, 2u * l + 1u, m->ff_dropout, stream));
  if (batched) {  // gated-GELU backward + FeedForward[0].bias partials in one pass (reduced below)
    CK(vbx_geglu_bwd_colsum(y.h1, a.dg, a.dh1, M, d.Fp, csl, stream));
  } else {
    CK(vbx_geglu_bwd(y.h1, a.dg, a.dh1, M, d.Fp, stream));
    CK(vbx_colsum_bf16(a.dh1, M, 2 * d.Fp, 2 * d.Fp, Gd + o[VBX_L_FF1B], 2 * d.F, 1, d.F, a.cs_scratch, stream));
  }
  { ProfScope ps("dgrad ff_in", st); CK(gemm_nn_bf16(a.dh1, 2 * d.Fp, w.layer[l].w1, d.D, M, d.D, 2 * d.Fp, a.dhn, d.D, st)); }
  CK(wgrad(a.dh1, 2 * d.Fp, y.hn2, d.D, 2 * d.Fp, d.D, d.M, a.slabs + sfl, Gd + o[VBX_L_FF1W], 2 * d.F, d.D, 1, d.F, st, wjp, wgp, gs, sq_at(1)));
  // (the column sums of the incoming dx -- FeedForward[3].bias gradient -- ride along in the same pass)
  CK(wgrad_join(st));  // the FeedForward-out wgrad reads a.dxb, which the norm backward below overwrites
  if (m->plain_norm) {
    CK(vbx_rmsnorm_bwd(x_mid, P + o[VBX_L_N2G], 0, a.dhn, a.dx, a.dx, dxb_attn, npart, cpart, d.B, d.Np, 0, d.Np, d.D, stream));
    if (!batched) {
      CK(vbx_reduce_norm_partials(npart, a.tscratch, 2 * d.D, d.B, chunks, d.D, 0, stream));
      CK(vbx_sum_rows_f32(a.tscratch, d.B, 2 * d.D, Gd + o[VBX_L_N2G], d.D, 0, stream));
    }
  } else {
    CK(vbx_rmsnorm_bwd(x_mid, ada_l + 2 * d.D, 4 * d.D, a.dhn, a.dx, a.dx, dxb_attn, npart, cpart, d.B, d.Np, 0, d.Np, d.D,
                       stream));
    if (!batched) CK(vbx_reduce_norm_partials(npart, dada_l + 2 * d.D, 4 * d.D, d.B, chunks, d.D, 0, stream));
  }
  if (!batched) CK(vbx_reduce_col_partials(cpart, Gd + o[VBX_L_FF2B], a.tscratch, d.B, chunks, d.D, stream));
  // ---- Attention
  // delta = rowsum(dO * O) of the attention backward rides in this GEMM's epilogue when the tile serving it has one (128 x 256 tile:
  // gemm_epi3.hpp::Epi3BF16Delta); otherwise the attention entry point runs its own pass over O and dO.
  // MEASURED (round 5, same box, in situ): the GEMM 21.4 -> 33.4 us per launch against 5 us saved in the attention stage -- the row
  // stage's 16 read-backs per wave each gain a dependent 16-byte load, three lane exchanges and a division on the critical path of a
  // tile that has only two workgroups per CU to hide them.  A loser by 7 us per layer: OFF by default, VBX_DELTA_FUSED=1 re-enables it.
  static const bool delta_fused = getenv("VBX_DELTA_FUSED") && atoi(getenv("VBX_DELTA_FUSED")) != 0;
  bool have_delta = false;
  {
    ProfScope ps("dgrad to_out", st);
    vbx_gemm_desc g{};
    g.mode = VBX_GEMM_NN; g.epilogue = VBX_EPI_BF16; g.M = M; g.N = d.I; g.K = d.D; g.lda = d.D; g.ldb = d.I; g.ldc = d.I;
    g.A = dxb_attn; g.B = w.layer[l].out; g.C = a.dO;
    int rc = VBX_EUNSUPPORTED;
    if (delta_fused && vbx_attn_bwd_variant() != 2) {
      g.delta_o = y.oh; g.delta = a.delta; g.H = d.H; g.Np = d.Np;
      rc = vbx_gemm(&g, st);
      have_delta = rc == 0;
    }
    if (rc == VBX_EUNSUPPORTED) {
      g.delta_o = nullptr; g.delta = nullptr;
      rc = vbx_gemm(&g, st);
    }
    CK(rc);
  }
  const u16* attn_out = have_delta ? nullptr : y.oh;  // NULL: a.delta is already there
  CK(wgrad(dxb_attn, d.D, y.o, d.I, d.D, d.I, d.M, a.slabs + 2 * sfl, Gd + o[VBX_L_OUTW], d.D, d.I, 0, 0, st, wjp, wgp, gs, sq_at(2)));
  static const bool fused_qk = !(getenv("VBX_ATTN_FUSED_QKBWD") && atoi(getenv("VBX_ATTN_FUSED_QKBWD")) == 0);  // 0: A/B
  if (fused_qk) {
    ProfScope ps("bwd attention", st);
    CK(vbx_attn_bwd_fused_dropout(y.q16, y.k16, y.qb, y.kb, y.v, io ? io->attn_mask_p : nullptr, attn_out, 1, a.dO, y.lse, a.delta, y.qrn,
                                  y.krn, m->qk_norm ? P + o[VBX_L_QG] : nullptr, m->qk_norm ? P + o[VBX_L_KG] : nullptr, m->rot_cos,
                                  m->rot_sin, m->qk_norm ? 8.0f : 0.0f, a.dqkv, 3 * d.I, gpart, d.B, d.H, d.Np, m->attn_scale,
                                  a.attn_scratch, drop_on ? y.dbr : nullptr, y.dbc, m->attn_dropout, stream));
    if (m->qk_norm && !batched) {
      const int rows = d.B * vbx_attn_bwd_fused_tiles(d.Np);
      CK(vbx_sum_rows_f32(gpart, rows, (long)d.H * 64, Gd + o[VBX_L_QG], (long)d.H * 64, 0, stream));
      CK(vbx_sum_rows_f32(gpart + (size_t)rows * d.H * 64, rows, (long)d.H * 64, Gd + o[VBX_L_KG], (long)d.H * 64, 0, stream));
    }
  } else {
    if (y.dbr && drop_on)
      CK(vbx_attn_bwd_dropout(y.q16, y.k16, y.qb, y.kb, y.v, io ? io->attn_mask_p : nullptr, attn_out, 1, a.dO, y.lse, a.delta, a.dq, a.dk,
                              a.dqkv + 2 * d.I, 3 * d.I, d.B, d.H, d.Np, m->attn_scale, y.dbr, y.dbc, m->attn_dropout, stream));
    else
      CK(vbx_attn_bwd(y.q16, y.k16, y.qb, y.kb, y.v, io ? io->attn_mask_p : nullptr, attn_out, 1, a.dO, y.lse, a.delta, a.dq, a.dk,
                      a.dqkv + 2 * d.I, 3 * d.I, d.B, d.H, d.Np, m->attn_scale, a.attn_scratch, stream));
    CK(vbx_qknorm_rope_bwd(a.dq, a.dk, y.q16, y.k16, y.qrn, y.krn, m->qk_norm ? P + o[VBX_L_QG] : nullptr,
                           m->qk_norm ? P + o[VBX_L_KG] : nullptr, m->rot_cos, m->rot_sin, m->qk_norm ? 8.0f : 0.0f, a.dqkv,
                           3 * d.I, gpart, d.B, d.H, d.Np, vbx_attn_q_prescale(m->attn_scale), stream));
    if (m->qk_norm) {
      const int rows = vbx_qknorm_rope_bwd_gpart_rows(d.B);
      CK(vbx_sum_rows_f32(gpart, rows, (long)d.H * 64, Gd + o[VBX_L_QG], (long)d.H * 64, 0, stream));
      CK(vbx_sum_rows_f32(gpart + (size_t)rows * d.H * 64, rows, (long)d.H * 64, Gd + o[VBX_L_KG], (long)d.H * 64, 0, stream));
    }
  }
  { ProfScope ps("dgrad to_qkv", st); CK(gemm_nn_bf16(a.dqkv, 3 * d.I, w.layer[l].qkv, d.D, M, d.D, 3 * d.I, a.dhn, d.D, st)); }
  CK(wgrad(a.dqkv, 3 * d.I, y.hn1, d.D, 3 * d.I, d.D, d.M, a.slabs + 3 * sfl, Gd + o[VBX_L_QKVW], 3 * d.I, d.D, 0, 0, st, wjp, wgp, gs, sq_at(3)));
  if (wgg.n) { ProfScope ps("wgrad (4 GEMMs)", st); CK(vbx_gemm_tn_splitk_grouped(wgg.d, wgg.n, stream)); }  // every operand is still live here (a.dxb: see dxb_attn)
  // VBX_LAYER_REDUCE=1: the slab reduction rides in the layer's batched reduce launch below (vbx_layer_reduce, bit-identical).
  // MEASURED (round 5, two interleaved runs per arm on one box): 9.95-10.02 vs 9.85-9.93 ms per step -- the 16 us of slab traffic now
  // sit behind the norm backward instead of overlapping the drain of the weight-gradient GEMM.  A loser: OFF by default.
  static const bool fuse_red_env = getenv("VBX_LAYER_REDUCE") && atoi(getenv("VBX_LAYER_REDUCE")) != 0;
  const bool fuse_red = fuse_red_env && batched && wj.n > 0 && !defer;
  if (wj.n && !fuse_red) { ProfScope ps("wgrad slab reduce", st); CK(vbx_splitk_reduce_multi(&wj, stream)); }
  CK(wgrad_join(st));  // the to_out wgrad reads a.dxb, which the norm backward below overwrites
  if (m->plain_norm) {
    CK(vbx_rmsnorm_bwd(x_in, P + o[VBX_L_N1G], 0, a.dhn, a.dx, a.dx, a.dxb, npart2, nullptr, d.B, d.Np, 0, d.Np, d.D, stream));
    if (!batched) {
      CK(vbx_reduce_norm_partials(npart2, a.tscratch, 2 * d.D, d.B, chunks, d.D, 0, stream));
      CK(vbx_sum_rows_f32(a.tscratch, d.B, 2 * d.D, Gd + o[VBX_L_N1G], d.D, 0, stream));
    }
  } else {
    CK(vbx_rmsnorm_bwd(x_in, ada_l, 4 * d.D, a.dhn, a.dx, a.dx, a.dxb, npart2, nullptr, d.B, d.Np, 0, d.Np, d.D, stream));
    if (!batched) CK(vbx_reduce_norm_partials(npart2, dada_l, 4 * d.D, d.B, chunks, d.D, 0, stream));
  }
  if (batched) {
    // ---- every small reduction of the layer in ONE launch (they cost ~0.5 ms per step as separate launches); deferred: layer 0
    // issues the jobs of ALL layers, six layers per launch
    vbx_mr_jobs jb{};
    auto add = [&](const float* src, float* dst, int rows, int cols, long row_stride, int batches, long sbs, long dbs, int dst_len,
                   int rowmap, int F) {
      vbx_mr_job& j = jb.job[jb.n++];
      j.src = src; j.dst = dst; j.rows = rows; j.cols = cols; j.row_stride = row_stride; j.batches = batches;
      j.src_bstride = sbs; j.dst_bstride = dbs; j.dst_len = dst_len; j.rowmap = rowmap; j.F = F;
    };
    const long rec = 2L * d.D;
    auto layer_jobs = [&](int ll, int region) {  // at most 8 jobs
      const long* ol = m->off + VBX_NG + (long)ll * VBX_NL;
      float* dada_ll = a.dada + (size_t)ll * d.B * 4 * d.D;
      const float* np1 = a.npart + region * a.np_stride;
      const float* np2 = a.npart2 + region * a.np_stride;
      const float* cp = a.cpart + region * a.cp_stride;
      const float* gp = a.gpart + region * a.gp_stride;
      const float* cs = defer ? a.cs_layers + region * a.cs_stride : a.cs_scratch;
      if (m->plain_norm) {  // d(gamma) = first half of the records, summed over batch and chunks
        add(np1, Gd + ol[VBX_L_N2G], d.B * chunks, d.D, rec, 1, 0, 0, d.D, 0, 0);
        add(np2, Gd + ol[VBX_L_N1G], d.B * chunks, d.D, rec, 1, 0, 0, d.D, 0, 0);
      } else {              // per-batch d(gamma | beta) of the two adaLN norms -> dada_l [B][g1 b1 g2 b2]
        add(np1, dada_ll + 2 * d.D, chunks, 2 * d.D, rec, d.B, (long)chunks * rec, 4L * d.D, 2 * d.D, 0, 0);
        add(np2, dada_ll, chunks, 2 * d.D, rec, d.B, (long)chunks * rec, 4L * d.D, 2 * d.D, 0, 0);
      }
      if (ada_all) {  // factor form: the projections' bias gradient sum_b dada[b][j] is the same records summed over batch as well
        add(np1, Gd + ol[VBX_L_G1B] + 2 * d.D, d.B * chunks, 2 * d.D, rec, 1, 0, 0, 2 * d.D, 0, 0);
        add(np2, Gd + ol[VBX_L_G1B], d.B * chunks, 2 * d.D, rec, 1, 0, 0, 2 * d.D, 0, 0);
      }
      add(cp, Gd + ol[VBX_L_FF2B], d.B * chunks, d.D, d.D, 1, 0, 0, d.D, 0, 0);                               // FeedForward[3].bias
      add(cs, Gd + ol[VBX_L_FF1B], vbx_geglu_bwd_colsum_slabs(), 2 * d.Fp, 2L * d.Fp, 1, 0, 0, 2 * d.F, 1, d.F);   // FeedForward[0].bias
      if (fused_qk && m->qk_norm) {
        const int rows = d.B * vbx_attn_bwd_fused_tiles(d.Np);
        add(gp, Gd + ol[VBX_L_QG], rows, d.H * 64, d.H * 64L, 1, 0, 0, d.H * 64, 0, 0);
        add(gp + (size_t)rows * d.H * 64, Gd + ol[VBX_L_KG], rows, d.H * 64, d.H * 64L, 1, 0, 0, d.H * 64, 0, 0);
      }
    };
    if (!defer) {
      layer_jobs(l, 0);
      if (fuse_red) { ProfScope ps("layer reduce (fused)", st); CK(vbx_layer_reduce(&wj, &jb, stream)); }
      else CK(vbx_multi_reduce(&jb, stream));
    } else if (l == 0) {
      ProfScope ps("partials reduce (all)", st);
      for (int ll = d.L - 1; ll >= 0; ll--) {
        layer_jobs(ll, ll);
        if (jb.n + 8 > VBX_MR_MAX || ll == 0) {
          CK(vbx_multi_reduce(&jb, stream));
          jb.n = 0;
        }
      }
    }
  }
  if (m->gateloop) {
    // ---- GateLoop: a.dx is the gradient of x_gl = LayerNorm(s) + x0; the residual branch stays in a.dx
    CK(vbx_layernorm_bwd(y.gls, P + o[VBX_L_GLLNW], a.dx, a.gl_ds, a.npart, d.B, d.Np, d.D, 1e-5f, stream));
    CK(vbx_reduce_norm_partials(a.npart, a.tscratch, 2 * d.D, d.B, ln_chunks, d.D, 0, stream));  // [B][dw|db]
    CK(vbx_sum_rows_f32(a.tscratch, d.B, 2 * d.D, Gd + o[VBX_L_GLLNW], 2 * d.D, 0, stream));
    CK(vbx_gateloop_scan_bwd(y.glp, y.glh, a.gl_ds, a.gl_dp, d.B, d.Np, d.D, stream));
    CK(gemm_nn_bf16(a.gl_dp, 3 * d.D, w.layer[l].glw, d.D, M, d.D, 3 * d.D, a.dhn, d.D, st));
    CK(wgrad(a.gl_dp, 3 * d.D, y.hg, d.D, 3 * d.D, d.D, d.M, a.slabs, Gd + o[VBX_L_GLW], 3 * d.D, d.D, 0, 0, st));
    CK(vbx_rmsnorm_bwd(layer_input(m, a, l), P + o[VBX_L_GLG], 0, a.dhn, a.dx, a.dx, a.dxb, a.npart, nullptr, d.B, d.Np, 0, d.Np, d.D, stream));
    CK(vbx_reduce_norm_partials(a.npart, a.tscratch, 2 * d.D, d.B, chunks, d.D, 0, stream));
    CK(vbx_sum_rows_f32(a.tscratch, d.B, 2 * d.D, Gd + o[VBX_L_GLG], d.D, 0, stream));
  }
  if (m->unet && l >= d.L / 2) {
    // ---- skip combiner: a.dx = d(combined input).  d(bias), dW = dx^T . cat, d(cat) = dx . W -> d(x) | d(skip)   (:458-463)
    const int p = d.L - 1 - l;
    CK(vbx_colsum_f32(a.dx, M, d.D, d.D, Gd + o[VBX_L_SKB], a.cs_scratch, stream));
    CK(vbx_unet_cat(a.xs[S * l], a.xs[S * p], m->skip_scale, nullptr, a.catb, d.M, d.D, stream));
    CK(wgrad(a.dxb, d.D, a.catb, 2 * d.D, d.D, 2 * d.D, d.M, a.slabs, Gd + o[VBX_L_SKW], d.D, 2 * d.D, 0, 0, st));
    vbx_gemm_desc g{};
    g.mode = VBX_GEMM_NN; g.epilogue = VBX_EPI_F32; g.M = M; g.N = 2 * d.D; g.K = d.D; g.lda = d.D; g.ldb = 2 * d.D; g.ldc = 2 * d.D;
    g.A = a.dxb; g.B = w.layer[l].skw; g.C = a.dcat;
    CK(vbx_gemm(&g, stream));
    CK(wgrad_join(st));  // the combiner's wgrad reads a.dxb, rewritten next
    CK(vbx_unet_split(a.dcat, m->skip_scale, a.dx, a.dxb, a.dskip[p], d.M, d.D, stream));
  } else if (m->unet) {
    // the input of a first-half layer also fed the combiner of layer L-1-l
    CK(vbx_unet_addskip(a.dx, a.dxb, a.dskip[l], d.M * d.D, stream));
  }
  if (m->plain_norm) return 0;
  // ---- this layer's adaLN projections (their 4 weights / 4 biases are contiguous): dW, dbias, and d(time_emb) +=
  // (adaln_factors: the weight gradient dada_l^T . temb is not materialised -- include/vbx.h, vbx_adam_adaln_factors)
  if (ada_all) {
    if (l == 0) {  // the layers run L-1 .. 0: every dada_l is in place
      ProfScope ps("adaLN dtemb (all)", st);
      CK(vbx_adaln_dtemb_all(w.adah, a.dada, a.dtemb, a.ada_scratch, d.L, d.B, d.Th, 4 * d.D, stream));
    }
    return 0;
  }
  CK(vbx_adaln_proj_bwd(m->stack_only ? io->cond : a.temb, w.adah + (size_t)l * 4 * d.D * d.Th, dada_l,
                        m->adaln_factors ? nullptr : Gd + o[VBX_L_G1W], Gd + o[VBX_L_G1B], a.dtemb,
                        a.ada_scratch, d.B, d.Th, 4 * d.D, l == d.L - 1 ? 0 : 1, stream));
  return 0;
}

static int backward_embed_impl(const vbx_model* m, const vbx_io* io, void* stream) {
  CK(check_model(m));
  VBX_REQUIRE(m->training && m->grads && io && (m->stack_only ? io->dx != nullptr : io->times != nullptr),
              "vbx_model_backward_embed: needs a training forward");
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  Acts a;
  carve_acts(m, a);
  const float* P = m->params;
  float* Gd = m->grads;
  const long* G = m->off;
  if (m->stack_only) {
    CK(vbx_stack_input_bwd(a.dx, io->dx, d.R ? Gd + G[VBX_P_REG] : nullptr, d.B, d.N, d.R, d.D, stream));
    if (!m->plain_norm && io->dcond &&
        hipMemcpyAsync(io->dcond, a.dtemb, (size_t)d.B * d.Th * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      vbx_set_error("vbx_model_backward_embed: copy of d(cond) failed");
      return VBX_EINVAL;
    }
    return 0;
  }
  CK(vbx_convpos_bwd(a.e, P + G[VBX_P_CONVW], P + G[VBX_P_CONVB], io->attn_mask, a.dx, a.dpre, a.de, a.deb, a.wpart,
                     d.R ? Gd + G[VBX_P_REG] : nullptr, d.B, d.N, d.R, d.D, d.ks, stream));
  CK(vbx_conv_wgrad_finalize(a.wpart, vbx_convpos_bwd_chunks(d.B, d.N), d.D, d.ks, Gd + G[VBX_P_CONVW], Gd + G[VBX_P_CONVB], stream));
  CK(wgrad(a.deb, d.D, a.embed_in, d.Ke, d.D, d.Ke, d.M0, a.slabs, Gd + G[VBX_P_EMBW], d.D, d.Ke, 0, 0, st));
  if (d.E) {
    // d(cond_emb) = de . W_embed[:, D:D+E]  ->  scatter into the embedding table gradient   (:1055, :1075-1076)
    WPack w;
    carve_wpack(m, w);
    CK(gemm_nn_bf16(a.deb, d.D, w.embb + d.Din, d.Ke, (int)d.M0, d.E, d.D, a.demb, d.E, st));
    if (hipMemsetAsync(Gd + G[VBX_P_CEMB], 0, (size_t)m->V1 * d.E * sizeof(float), st) != hipSuccess) {
      vbx_set_error("vbx_model_backward_embed: memset of the embedding gradient failed");
      return VBX_EINVAL;
    }
    CK(vbx_cond_emb_bwd(a.demb, d.E, io->cond_ids, io->T, io->drop_mask, io->null_id, Gd + G[VBX_P_CEMB], d.B, d.N, d.E, stream));
  }
  CK(vbx_colsum_f32(a.de, (int)d.M0, d.D, d.D, Gd + G[VBX_P_EMBB], a.cs_scratch, stream));
  CK(vbx_time_embed_bwd(io->times, P + G[VBX_P_SINW], P + G[VBX_P_T1W], a.four, a.pre, a.dtemb, Gd + G[VBX_P_SINW],
                        Gd + G[VBX_P_T1W], Gd + G[VBX_P_T1B], a.tscratch, d.B, d.D, d.Th, stream));
  return 0;
}

// Public backward stages: each returns with every gradient of the stage ordered before later work on the caller's stream
// (weight gradients run on the side stream, see wgrad()).
extern "C" int vbx_model_backward_head(const vbx_model* m, const vbx_io* io, const float* gscale, void* stream) {
  const int rc = backward_head_impl(m, io, gscale, stream);
  const int rj = wgrad_join((hipStream_t)stream);
  return rc ? rc : rj;
}
extern "C" int vbx_model_backward_layer(const vbx_model* m, const vbx_io* io, int l, void* stream) {
  const int rc = backward_layer_impl(m, io, l, stream);
  const int rj = wgrad_join((hipStream_t)stream);
  return rc ? rc : rj;
}
extern "C" int vbx_model_backward_embed(const vbx_model* m, const vbx_io* io, void* stream) {
  const int rc = backward_embed_impl(m, io, stream);
  const int rj = wgrad_join((hipStream_t)stream);
  return rc ? rc : rj;
}

// Segment table of the fused Adam + repack step (vbx_adam_step_packed): every parameter that vbx_model_pack_weights
// copies into the operand arena, in flat-buffer order, with plain segments in between.
extern "C" int vbx_model_adam_segments(const vbx_model* m, long n_flat, vbx_adam_seg* out, int max_segs, long* total_blocks) {
  CK(check_model(m));
  VBX_REQUIRE(n_flat > 0 && total_blocks, "vbx_model_adam_segments: bad args");
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  const long* G = m->off;
  std::vector<vbx_adam_seg> ps;
  auto add = [&](long off, long rows, long cols, void* b, void* h16, float* f32, int ld, int rowmap, int F) {
    vbx_adam_seg s{};
    s.off = off; s.count = rows * cols; s.dst_bf16 = b; s.dst_f16 = h16; s.dst_f32 = f32; s.cols = (int)cols; s.dst_ld = ld;
    s.rowmap = rowmap; s.F = F;
    ps.push_back(s);
  };
  if (!m->stack_only) {
    add(G[VBX_P_EMBW], d.D, d.Ke, w.embb, w.embh, nullptr, d.Ke, 0, 0);
    add(G[VBX_P_PREDW], d.Din, d.D, w.pred, w.predh, nullptr, d.D, 0, 0);
  }
  for (int l = 0; l < d.L; l++) {
    const long* o = m->off + VBX_NG + (long)l * VBX_NL;
    if (!m->plain_norm) {
      // (factor mode: the block is served by vbx_adam_adaln_factors -- rowmap 2 marks it, it gets no blocks of the fused launch)
      add(o[VBX_L_G1W], 4 * d.D, d.Th, nullptr, w.adah + (size_t)l * 4 * d.D * d.Th, nullptr, d.Th, m->adaln_factors ? 2 : 0, 0);
      add(o[VBX_L_G1B], 1, 4 * d.D, nullptr, nullptr, w.bada + (size_t)l * 4 * d.D, 4 * d.D, 0, 0);
    }
    add(o[VBX_L_QKVW], 3 * d.I, d.D, w.layer[l].qkv, w.layer[l].qkvh, nullptr, d.D, 0, 0);
    add(o[VBX_L_OUTW], d.D, d.I, w.layer[l].out, w.layer[l].outh, nullptr, d.I, 0, 0);
    add(o[VBX_L_FF1W], 2 * d.F, d.D, w.layer[l].w1, w.layer[l].w1h, nullptr, d.D, 1, d.F);
    add(o[VBX_L_FF1B], 2 * d.F, 1, nullptr, nullptr, w.layer[l].b1, 1, 1, d.F);
    add(o[VBX_L_FF2W], d.D, d.F, w.layer[l].w2, w.layer[l].w2h, nullptr, d.Fp, 0, 0);
    if (m->gateloop) add(o[VBX_L_GLW], 3 * d.D, d.D, w.layer[l].glw, w.layer[l].glwh, nullptr, d.D, 0, 0);
    if (w.layer[l].skw) add(o[VBX_L_SKW], d.D, 2 * d.D, w.layer[l].skw, w.layer[l].skwh, nullptr, 2 * d.D, 0, 0);
  }
  std::sort(ps.begin(), ps.end(), [](const vbx_adam_seg& a, const vbx_adam_seg& b) { return a.off < b.off; });
  std::vector<vbx_adam_seg> all;
  long cur = 0;
  for (const auto& s : ps) {
    VBX_REQUIRE(s.off >= cur && s.off + s.count <= n_flat, "vbx_model_adam_segments: overlapping / out-of-range parameter slots");
    if (s.off > cur) { vbx_adam_seg gseg{}; gseg.off = cur; gseg.count = s.off - cur; gseg.cols = 1; all.push_back(gseg); }
    all.push_back(s);
    cur = s.off + s.count;
  }
  if (cur < n_flat) { vbx_adam_seg gseg{}; gseg.off = cur; gseg.count = n_flat - cur; gseg.cols = 1; all.push_back(gseg); }
  long blocks = 0;
  for (auto& s : all) { s.block0 = blocks; blocks += s.rowmap == 2 ? 0 : (s.count + 2047) / 2048; }
  *total_blocks = blocks;
  if (out) {
    VBX_REQUIRE((int)all.size() <= max_segs, "vbx_model_adam_segments: table needs %d entries", (int)all.size());
    for (size_t i = 0; i < all.size(); i++) out[i] = all[i];
  }
  return (int)all.size();
}

extern "C" int vbx_model_adaln_factors(const vbx_model* m, const float** dada, const float** temb, long* w_off, void** dst_f16) {
  CK(check_model(m));
  VBX_REQUIRE(m->training && !m->plain_norm && !m->stack_only, "vbx_model_adaln_factors: needs a training VoiceBox arena with adaptive norms");
  const Dims d = dims_of(m);
  WPack w;
  carve_wpack(m, w);
  Acts a;
  carve_acts(m, a);
  if (dada) *dada = a.dada;
  if (temb) *temb = a.temb;
  for (int l = 0; l < d.L; l++) {
    if (w_off) w_off[l] = (m->off + VBX_NG + (long)l * VBX_NL)[VBX_L_G1W];
    if (dst_f16) dst_f16[l] = w.adah + (size_t)l * 4 * d.D * d.Th;
  }
  return d.L;
}

extern "C" int vbx_prof_enable(int on) {
  for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.recs.clear();
  g_prof.on = on != 0;
  return 0;
}
extern "C" int vbx_prof_collect(vbx_prof_entry* out, int max_entries) {
  VBX_REQUIRE(out && max_entries > 0, "vbx_prof_collect: bad args");
  g_prof.on = false;
  int n = 0;
  for (auto& r : g_prof.recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    int k = 0;
    for (; k < n; k++)
      if (!strncmp(out[k].label, r.label, sizeof(out[k].label) - 1)) break;
    if (k == n) {
      if (n == max_entries) continue;
      memset(&out[n], 0, sizeof(out[n]));
      strncpy(out[n].label, r.label, sizeof(out[n].label) - 1);
      n++;
    }
    out[k].calls++;
    out[k].total_us += ms * 1000.f;
  }
  for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.recs.clear();
  return n;
}

// Debug/introspection (tests only): device pointer of a named arena tensor.
extern "C" void* vbx_model_debug_ptr(const vbx_model* m, const char* name, int layer) {
  Acts a;
  carve_acts(m, a);
  const std::string n(name);
  const Dims d = dims_of(m);
  if (layer >= 0 && layer < d.L) {
    const ALayer& y = a.layer[layer];
    if (n == "hn1") return y.hn1; if (n == "hn1h") return y.hn1h; if (n == "q16") return y.q16; if (n == "k16") return y.k16;
    if (n == "qb") return y.qb; if (n == "kb") return y.kb; if (n == "v") return y.v; if (n == "vh") return y.vh;
    if (n == "o") return y.o; if (n == "oh") return y.oh; if (n == "lse") return y.lse; if (n == "qrn") return y.qrn;
    if (n == "glp") return y.glp; if (n == "glh") return y.glh; if (n == "gls") return y.gls;
    if (n == "krn") return y.krn; if (n == "hn2") return y.hn2; if (n == "h1") return y.h1; if (n == "g") return y.g;
  }
  if (n == "xs" && layer >= 0 && layer < (int)a.xs.size()) return a.xs[layer];
  if (n == "dx") return a.dx; if (n == "dxb") return a.dxb; if (n == "dq") return a.dq; if (n == "dk") return a.dk;
  if (n == "dqkv") return a.dqkv; if (n == "dO") return a.dO; if (n == "delta") return a.delta; if (n == "dhn") return a.dhn;
  if (n == "e") return a.e; if (n == "temb") return a.temb; if (n == "ada") return a.ada; if (n == "pred") return a.pred;
  if (n == "dada") return a.dada; if (n == "dtemb") return a.dtemb;
  return nullptr;
}
