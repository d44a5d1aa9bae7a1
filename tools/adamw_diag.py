"""Diagnostic: where do the autograd path (wrapper(x).backward()) and TrainStep.step differ on the dim-64 golden?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voicebox_pytorch_amd as vbx
from voicebox_pytorch_amd.dp import TrainStep
from voicebox_pytorch_amd.masks import rng_override
dev = "cuda"
g = torch.load("tests/golden/small.pt", map_location="cpu", weights_only=False)
draws = dict(x0=g["x0"], times=g["times"], frac_lengths=g["frac"], rand=g["rand"])
def make():
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=500, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(g["state"], strict=False)
    vb = vb.to(dev)
    return vb, vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
vb_r, w_r = make()
vb, w = make()
ts = TrainStep(w, lr=1e-3, max_grad_norm=0.5)
with rng_override(**draws):
    l_r = w_r(g["x1"].to(dev)); l_r.backward()
with rng_override(**draws):
    l = ts._forward_backward(g["x1"].to(dev), None, None, None)
print("loss autograd", float(l_r), "trainstep", float(l))
off = ts.fp.offsets if hasattr(ts.fp, "offsets") else None
names = dict(vb_r.named_parameters())
views = ts.fp.views(ts.gflat) if hasattr(ts.fp, "views") else None
worst = []
for (k, p), (k2, p2) in zip(vb_r.named_parameters(), vb.named_parameters()):
    if p.grad is None: continue
    gts = ts.fp.grad_view(ts.gflat, k2) if hasattr(ts.fp, "grad_view") else None
    if gts is None: break
    e = float((p.grad - gts).norm() / p.grad.norm().clamp(min=1e-30))
    worst.append((e, k, float(p.grad.norm())))
print(sorted(worst, reverse=True)[:8])
print([a for a in dir(ts.fp) if not a.startswith("_")])
