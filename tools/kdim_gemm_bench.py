"""Stand-alone timing of the two K = dim forward GEMMs with their real fused epilogues at the benchmark shape (dim 512, 16 heads,
B = 8 x 1040 rows): to_qkv (+ MultiheadRMSNorm + rotary + head split) and FeedForward-in (+ GEGLU), each as the training step
launches it (fp16 + bf16 outputs, saved pre-activation) and as the sampler does (fp16 outputs only).  Descriptors mirror
csrc/runtime.hip.  Usage: python tools/kdim_gemm_bench.py [iters]   (VBX_GEMM_PATH / VBX_GEMM_ABL apply)."""
import math, os, sys, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, Np, D, H = int(os.environ.get("BATCH", 8)), int(os.environ.get("NP", 1040)), int(os.environ.get("DIM", 512)), 16
I = H * 64
M = B * Np
F = int(D * 4 * 2 / 3)
Fp = (F + 63) // 64 * 64
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
lib = L.lib()


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


x = torch.randn(M, D, generator=g).half().to(dev)
Wqkv = (torch.randn(3 * I, D, generator=g) * D ** -0.5).half().to(dev)
qg = torch.ones(H, 64, device=dev); kg = torch.ones(H, 64, device=dev)
pos = torch.arange(Np).float()
inv = 1.0 / (50000.0 ** (torch.arange(0, 64, 2).float() / 64))
fr = pos[:, None] * inv[None, :]
rc, rs = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
q16 = torch.empty(B, H, Np, 64, dtype=torch.float16, device=dev); k16 = torch.empty_like(q16); v16 = torch.empty_like(q16)
qb = torch.empty(B, H, Np, 64, dtype=torch.bfloat16, device=dev); kb = torch.empty_like(qb); vb = torch.empty_like(qb)
qrn = torch.empty(B, H, Np, device=dev); krn = torch.empty_like(qrn)


def qkv(train):
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb = L.VBX_GEMM_NT, L.VBX_EPI_QKV, M, 3 * I, D, D, D
    d.A, d.B, d.f16, d.Np, d.H, d.qk_scale = x.data_ptr(), Wqkv.data_ptr(), 1, Np, H, 8.0
    d.q_gamma, d.k_gamma, d.rot_cos, d.rot_sin = qg.data_ptr(), kg.data_ptr(), rc.data_ptr(), rs.data_ptr()
    d.q16, d.k16, d.v16 = q16.data_ptr(), k16.data_ptr(), v16.data_ptr()
    if train:
        d.qb, d.kb, d.v, d.q_rnorm, d.k_rnorm = qb.data_ptr(), kb.data_ptr(), vb.data_ptr(), qrn.data_ptr(), krn.data_ptr()
    d.q_prescale = lib.vbx_attn_q_prescale(10.0)
    return d


W1 = (torch.randn(2 * Fp, D, generator=g) * D ** -0.5).half().to(dev)
b1 = torch.zeros(2 * Fp, device=dev)
gh = torch.empty(M, Fp, dtype=torch.float16, device=dev); gb = torch.empty(M, Fp, dtype=torch.bfloat16, device=dev)
h1 = torch.empty(M, 2 * Fp, dtype=torch.bfloat16, device=dev)


def ffin(train):
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb = L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, M, 2 * Fp, D, D, D
    d.A, d.B, d.f16, d.C, d.ldc, d.bias = x.data_ptr(), W1.data_ptr(), 1, gh.data_ptr(), Fp, b1.data_ptr()
    if train:
        d.C2, d.C3 = h1.data_ptr(), gb.data_ptr()
    return d


cases = (("to_qkv train", qkv(True), 2.0 * M * 3 * I * D), ("to_qkv infer", qkv(False), 2.0 * M * 3 * I * D),
         ("ff_in  train", ffin(True), 2.0 * M * 2 * Fp * D), ("ff_in  infer", ffin(False), 2.0 * M * 2 * Fp * D))
paths = [int(x) for x in os.environ.get("PATHS", "0").split(",")]  # vbx_gemm_select values, interleaved over ROUNDS rounds in one process
rounds = int(os.environ.get("ROUNDS", 1))
res = {}
for r in range(rounds):
    for pth in paths:
        lib.vbx_gemm_select(pth)
        for name, d, fl in cases:
            def run(d=d):
                rc_ = lib.vbx_gemm(d, st)
                assert rc_ == 0, lib.vbx_last_error()
            res.setdefault((pth, name), []).append(timeit(run))
lib.vbx_gemm_select(0)
for pth in paths:
    for name, d, fl in cases:
        t = sorted(res[(pth, name)])
        us = t[len(t) // 2]
        print(f"path {pth} {name}: median {us:6.1f} us (min {t[0]:.1f}, max {t[-1]:.1f})  {fl / us / 1e6:6.0f} TFLOP/s ({fl / us / 1e6 / 25:.1f} % of 2.5 PF)")
