"""Stand-alone timing of the row-norm kernels at the benchmark shape (B = 8, Np = 1040, D = 512) with rotating buffers (12 layers' worth,
so the streams come from HBM / MALL as in the step): vbx_rmsnorm_fwd (training: bf16 + fp16 out; eval: fp16) and vbx_rmsnorm_bwd (with
the incoming residual gradient, bf16 copy, gamma / beta partials and the fused column sums).  Usage: python tools/norm_bench.py"""
import os, sys, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = "cuda"
st = L.current_stream
B, Np, D = 8, int(os.environ.get("NP", 1040)), int(os.environ.get("DIM", 512))
NBUF = 12
g = torch.Generator().manual_seed(0)
xs = [torch.randn(B, Np, D, generator=g).to(dev) for _ in range(NBUF)]
ada = torch.randn(B, 4 * D, generator=g).to(dev)
yb = [torch.empty(B, Np, D, dtype=torch.bfloat16, device=dev) for _ in range(NBUF)]
yh = [torch.empty(B, Np, D, dtype=torch.float16, device=dev) for _ in range(NBUF)]
dys = [(torch.randn(B, Np, D, generator=g) * 1e-3).bfloat16().to(dev) for _ in range(NBUF)]
dxin = [torch.randn(B, Np, D, generator=g).to(dev) * 1e-3 for _ in range(NBUF)]
dxo = [torch.empty(B, Np, D, device=dev) for _ in range(NBUF)]
dxb = [torch.empty(B, Np, D, dtype=torch.bfloat16, device=dev) for _ in range(NBUF)]
chunks = L.lib().vbx_rmsnorm_bwd_chunks(Np)
part = torch.empty(B, chunks, 2, D, device=dev)
cpart = torch.empty(B, chunks, D, device=dev)
i = [0]


def timeit(fn, iters=240):
    for _ in range(24):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def fwd_train():
    k = i[0] = (i[0] + 1) % NBUF
    L.call("vbx_rmsnorm_fwd", xs[k], ada, ada[:, D:], 4 * D, yb[k], yh[k], B, Np, 0, Np, D, st())


def fwd_eval():
    k = i[0] = (i[0] + 1) % NBUF
    L.call("vbx_rmsnorm_fwd", xs[k], ada, ada[:, D:], 4 * D, None, yh[k], B, Np, 0, Np, D, st())


def bwd():
    k = i[0] = (i[0] + 1) % NBUF
    L.call("vbx_rmsnorm_bwd", xs[k], ada, 4 * D, dys[k], dxin[k], dxo[k], dxb[k], part, cpart, B, Np, 0, Np, D, st())


n = B * Np * D
for name, fn, nbytes in (("rmsnorm_fwd train", fwd_train, n * (4 + 2 + 2)), ("rmsnorm_fwd eval", fwd_eval, n * (4 + 2)),
                         ("rmsnorm_bwd", bwd, n * (4 + 2 + 4 + 4 + 2))):
    ts = sorted(timeit(fn) for _ in range(5))
    us = ts[2]
    print(f"{name:18s} median {us:6.2f} us (min {ts[0]:.2f})  {nbytes / us / 1e6:5.2f} TB/s")
