"""Attention forward time vs sequence length around the benchmark's Np = 1040 = 8 full 128-row query tiles + 16 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.gemm_bench import timeit, L, st, dev
B, H = 8, 16
for Np in (896, 1024, 1040, 1056, 1152):
    q = torch.randn(B, H, Np, 64, device=dev); q = (q / q.norm(dim=-1, keepdim=True) * 8 * L.lib().vbx_attn_q_prescale(10.0)).half()  # q16 contract: include/vbx.h
    k = torch.randn(B, H, Np, 64, device=dev); k = (k / k.norm(dim=-1, keepdim=True) * 8).half()
    v = torch.randn(B, H, Np, 64, device=dev).half()
    out = torch.empty(B, Np, H * 64, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, Np, device=dev)
    sec = timeit(lambda: L.call("vbx_attn_fwd", q, k, v, None, out, None, lse, B, H, Np, 10.0, st))
    fl = 4.0 * B * H * Np * Np * 64
    print(f"Np={Np:5d}  q-tiles/head {(Np+127)//128}  workgroups {B*H*((Np+127)//128):5d}  {sec*1e6:7.1f} us  {fl/sec/1e12:6.1f} TF/s")
