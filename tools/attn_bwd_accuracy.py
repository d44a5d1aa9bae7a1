"""Accuracy of the attention backward kernels against an fp64 reference on the same (rounded) operands, at the reference's logit scale
(|q| = |k| = 8, scale 10: near-one-hot softmax): the folded bodies (select 1, round 5) against round 3's unfolded bodies (select 3).
Usage (GPU box): python tools/attn_bwd_accuracy.py"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voicebox_pytorch_amd import _lib as L
from oracle import restate
dev = "cuda"
st = lambda: torch.cuda.current_stream().cuda_stream
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for (B, H, Np, seed) in ((2, 2, 1040, 1), (2, 2, 1040, 2), (2, 2, 203, 3), (1, 4, 520, 4), (2, 2, 96, 5)):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, H, Np, 64, generator=g); k = torch.randn(B, H, Np, 64, generator=g); v = torch.randn(B, H, Np, 64, generator=g).half()
    q = (q / q.norm(dim=-1, keepdim=True) * 8).half(); k = (k / k.norm(dim=-1, keepdim=True) * 8).half()
    c = L.lib().vbx_attn_q_prescale(10.0)
    qs = (q.float() * c).half(); q_eff = qs.double() / c
    out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev); lse = torch.empty(B, H, Np, device=dev)
    L.call("vbx_attn_fwd", qs.to(dev), k.to(dev), v.to(dev), None, out16, None, lse, B, H, Np, 10.0, st())
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q_eff, k, v))
    ref = restate.attend(qr, kr, vr, scale=10.0).permute(0, 2, 1, 3).reshape(B, Np, H * 64)
    dout = (torch.randn(B, Np, H * 64, generator=g) * 1e-3).bfloat16()
    ref.backward(dout.double())
    res = {}
    for variant in (1, 3):
        L.lib().vbx_attn_bwd_select(variant)
        delta = torch.empty(B, H, Np, device=dev); dq = torch.zeros(B, H, Np, 64, device=dev); dk = torch.zeros_like(dq)
        dv = torch.zeros(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
        L.call("vbx_attn_bwd", qs.to(dev), k.to(dev), q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev), None, out16, 1,
               dout.to(dev), lse, delta, dq, dk, dv.data_ptr(), H * 64, B, H, Np, 10.0, None, st())
        torch.cuda.synchronize()
        dvh = dv.float().cpu().view(B, Np, H, 64).permute(0, 2, 1, 3)
        res[variant] = (rel(dq, qr.grad), rel(dk, kr.grad), rel(dvh, vr.grad), dq.clone(), dk.clone())
    L.lib().vbx_attn_bwd_select(0)
    print(f"B{B} H{H} Np{Np}: folded dq {res[1][0]:.4f} dk {res[1][1]:.4f} dv {res[1][2]:.4f} | unfolded dq {res[3][0]:.4f} dk {res[3][1]:.4f} dv {res[3][2]:.4f}"
          f" | folded vs unfolded dq {rel(res[1][3], res[3][3]):.4f} dk {rel(res[1][4], res[3][4]):.4f}")
