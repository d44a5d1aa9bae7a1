"""Stand-alone attention forward/backward timing at the benchmark shape (B=8, H=16, Np=1040 = 1024 frames + 16 register tokens; NP=<n> overrides, |q|=|k|=8, scale 10).
Usage: python tools/attn_bench.py [iters]   (run under rocprofv3 --kernel-trace --stats for per-kernel durations)."""
import os, sys, torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
B, H, Np = int(os.environ.get("BATCH", 8)), 16, int(os.environ.get("NP", 1040))  # frames + 16 register tokens
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"  # evict L2 + MALL between launches (per-kernel times then come from rocprofv3)
scrub = torch.empty(1 << 28, dtype=torch.float32, device="cuda:0") if cold else None
g = torch.Generator().manual_seed(0)
q = torch.randn(B, H, Np, 64, generator=g); k = torch.randn(B, H, Np, 64, generator=g); v = torch.randn(B, H, Np, 64, generator=g)
q = q / q.norm(dim=-1, keepdim=True) * 8; k = k / k.norm(dim=-1, keepdim=True) * 8
qd, kd, vd = (q * L.lib().vbx_attn_q_prescale(10.0)).half().to(dev), k.half().to(dev), v.half().to(dev)  # q16 contract: include/vbx.h
qb, kb, vb = q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev)
out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev)
out = torch.empty(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, Np, device=dev)
st = torch.cuda.current_stream().cuda_stream
dout = (torch.randn(B, Np, H * 64, generator=g) * 1e-3).bfloat16().to(dev)
delta = torch.empty(B, H, Np, device=dev)
dq = torch.zeros(B, H, Np, 64, device=dev); dk = torch.zeros(B, H, Np, 64, device=dev)
dv = torch.zeros(B, Np, H * 64, dtype=torch.bfloat16, device=dev)


scratch = None  # (no kernel needs scratch since round 6)


def fwd():
    L.call("vbx_attn_fwd", qd, kd, vd, None, out16, out, lse, B, H, Np, 10.0, st)


def fwd_eval():  # inference: no bf16 copy of the output
    L.call("vbx_attn_fwd", qd, kd, vd, None, out16, None, lse, B, H, Np, 10.0, st)


def bwd():
    L.call("vbx_attn_bwd", qd, kd, qb, kb, vb, None, out16, 1, dout, lse, delta, dq, dk, dv.data_ptr(), H * 64, B, H, Np, 10.0, scratch, st)


for fn, name in ((fwd, "fwd"), (fwd_eval, "fwd_eval"), (bwd, "bwd")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        if cold:
            scrub.add_(1.0)
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / iters * 1e3:.1f} us")
