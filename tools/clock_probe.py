"""Shader clock during the attention forward kernel (VBX_ATTN_ABL2=64 build of the kernel writes cycles / 100 MHz ticks)."""
import os, sys
os.environ["VBX_ATTN_ABL2"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicebox_pytorch_amd import _lib as L
dev = "cuda"
st = L.current_stream()
B, H, Np = 8, 16, 1040
q = torch.randn(B, H, Np, 64, device=dev); q = (q / q.norm(dim=-1, keepdim=True) * 8 * L.lib().vbx_attn_q_prescale(10.0)).half()  # q16 contract: include/vbx.h
k = torch.randn(B, H, Np, 64, device=dev); k = (k / k.norm(dim=-1, keepdim=True) * 8).half()
v = torch.randn(B, H, Np, 64, device=dev).half()
out = torch.empty(B, Np, H * 64, device=dev, dtype=torch.float16)
lse = torch.empty(B, H, Np, device=dev)
for i in range(30):
    L.call("vbx_attn_fwd", q, k, v, None, out, None, lse, B, H, Np, 10.0, st)
torch.cuda.synchronize()
c = lse[:, :, 0].flatten().cpu(); w = lse[:, :, 1].flatten().cpu()
mhz = c / w * 100.0
print(f"workgroup lifetime: {c.mean():.0f} cycles, {w.mean()/100:.2f} us; shader clock {mhz.mean():.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
