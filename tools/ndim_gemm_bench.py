"""Stand-alone timing of the four N = dim GEMMs of a layer at the benchmark shape (M = 8 x 1040 rows; BATCH / NP override): fwd to_out
(NT fp16, K = 1024, fp32 out + residual), fwd FeedForward-out (NT fp16, K = 1408, + bias + residual), dgrad to_qkv (NN bf16, K = 3072),
dgrad FeedForward-in (NN bf16, K = 2816); PATHS=0,1,... interleaves vbx_gemm_select values in one process.
(Round 6 used it for the stream-K experiment recorded in DESIGN.md section 8: 208 tiles cut into 256 equal (tile, k-tile) shares whose
pieces meet through a workspace -- correct, deterministic, and 46-58 % SLOWER at these sizes: the publish / acquire seam costs more than
the 19 % of idle CUs it fills.)   Usage: python tools/ndim_gemm_bench.py [iters]"""
import os, sys, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, Np, D = int(os.environ.get("BATCH", 8)), int(os.environ.get("NP", 1040)), int(os.environ.get("DIM", 512))
M = B * Np
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rounds = int(os.environ.get("ROUNDS", 5))
g = torch.Generator().manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
lib = L.lib()
Fp = (int(D * 4 * 2 / 3) + 63) // 64 * 64


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(name, mode, K, N=None):
    N = N or D
    keep = []
    d = L.GemmDesc()
    if mode == "nt":
        A = torch.randn(M, K, generator=g).half().to(dev); Bw = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
        C = torch.empty(M, N, device=dev); bias = torch.zeros(N, device=dev); resid = torch.randn(M, N, generator=g).to(dev)
        d.mode, d.epilogue, d.f16, d.lda, d.ldb = L.VBX_GEMM_NT, L.VBX_EPI_F32, 1, K, K
        d.bias, d.resid = bias.data_ptr(), resid.data_ptr()
        keep += [bias, resid]
    else:
        A = torch.randn(M, K, generator=g).bfloat16().to(dev); Bw = (torch.randn(K, N, generator=g) * K ** -0.5).bfloat16().to(dev)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        d.mode, d.epilogue, d.lda, d.ldb = L.VBX_GEMM_NN, L.VBX_EPI_BF16, K, N
    d.M, d.N, d.K, d.ldc, d.A, d.B, d.C = M, N, K, N, A.data_ptr(), Bw.data_ptr(), C.data_ptr()
    keep += [A, Bw, C]
    return name, d, 2.0 * M * N * K, keep


cases = [case("fwd to_out      (NT K=1024)", "nt", 2 * D), case("fwd ff_out      (NT K=%d)" % Fp, "nt", Fp),
         case("dgrad to_qkv    (NN K=3072)", "nn", 6 * D), case("dgrad ff_in     (NN K=%d)" % (2 * Fp), "nn", 2 * Fp)]
paths = [int(x) for x in os.environ.get("PATHS", "0").split(",")]
res = {}
for r in range(rounds):
    for pth in paths:
        lib.vbx_gemm_select(pth)
        for name, d, fl, _ in cases:
            def run(d=d):
                assert lib.vbx_gemm(d, st) == 0, lib.vbx_last_error()
            res.setdefault((name, pth), []).append(timeit(run))
lib.vbx_gemm_select(0)
for pth in paths:
    tot = 0.0
    for name, d, fl, _ in cases:
        a = sorted(res[(name, pth)])
        ma = a[len(a) // 2]
        tot += ma
        print(f"path {pth} {name}: median {ma:6.1f} us (min {a[0]:.1f})  {fl / ma / 1e6:5.0f} TF/s")
    print(f"path {pth} sum: {tot:.1f} us")
