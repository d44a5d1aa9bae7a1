"""Stand-alone timing of single C-ABI ops at the bench shape (dim 512, B 8, Np 1040, Th 2048) -- a proxy for memory-bound kernels
only (GEMMs/attention must be judged inside the step).  usage: [VBX_LIB_PATH=...] python tools/op_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = "cuda"
st = L.current_stream


def timeit(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


B, Np, D, Th = 8, 1040, 512, 2048
J = 4 * D
temb = torch.randn(B, Th, device=dev)
W = torch.randn(J, Th, device=dev).half()
dada = torch.randn(B, J, device=dev)
dW, dbias, dtemb = torch.empty(J, Th, device=dev), torch.empty(J, device=dev), torch.empty(B, Th, device=dev)
scratch = torch.empty(L.lib().vbx_adaln_proj_bwd_scratch_floats(B, Th, J), device=dev)
# rotate over several buffers so that the weight stream comes from HBM, as in the step
Ws = [W.clone() for _ in range(12)]
dWs = [torch.empty_like(dW) for _ in range(12)]
i = [0]


def adaln_bwd():
    k = i[0] = (i[0] + 1) % 12
    L.call("vbx_adaln_proj_bwd", temb, Ws[k], dada, dWs[k], dbias, dtemb, scratch, B, Th, J, 0, st())


print("adaln_proj_bwd (kernel + sum_rows) us:", round(timeit(adaln_bwd), 2))

x = torch.randn(B, Np, D, device=dev)
gamma = torch.randn(B, D, device=dev)
dy = torch.randn(B * Np, D, device=dev).bfloat16()
dxin = torch.randn(B, Np, D, device=dev)
dxo = torch.empty_like(x)
dxb = torch.empty(B, Np, D, device=dev, dtype=torch.bfloat16)
chunks = L.lib().vbx_rmsnorm_bwd_chunks(Np)
part = torch.empty(B * chunks * 2 * D, device=dev)
cpart = torch.empty(B * chunks * D, device=dev)


def rms_bwd():
    L.call("vbx_rmsnorm_bwd", x, gamma, D, dy, dxin, dxo, dxb, part, cpart, B, Np, 0, Np, D, st())


print("rmsnorm_bwd us:", round(timeit(rms_bwd), 2), "chunks", chunks)
y = torch.empty(B * Np, D, device=dev, dtype=torch.bfloat16)
y16 = torch.empty(B * Np, D, device=dev, dtype=torch.float16)
beta = torch.randn(B, D, device=dev)


def rms_fwd():
    L.call("vbx_rmsnorm_fwd", x, gamma, beta, D, y, y16, B, Np, 0, Np, D, st())


print("rmsnorm_fwd us:", round(timeit(rms_fwd), 2))
