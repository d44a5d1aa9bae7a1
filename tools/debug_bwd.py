"""Debug helper (GPU box): stage-by-stage backward of the small golden model vs the fp64 oracle."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restate as R
import voicebox_pytorch_amd as vbx
from voicebox_pytorch_amd import engine as E, _lib
from voicebox_pytorch_amd.masks import rng_override

g = torch.load('tests/golden/small.pt', weights_only=False)
cfg = R.Cfg(**g['cfg'])
dev = 'cuda'
# ---- oracle with captured attention intermediates
cap = []
orig = R.attend
def hook(q, k, v, mask=None, scale=None):
    for t in (q, k, v): t.retain_grad()
    o = orig(q, k, v, mask, scale); o.retain_grad(); cap.append((q, k, v, o)); return o
R.attend = hook
pp = {k: v.double().clone().requires_grad_(v.is_floating_point() and k != 'null_cond') for k, v in g['state'].items()}
loss = R.cfm_loss(pp, cfg, g['x1'].double(), g['x0'].double(), g['times'].double(), g['frac'], g['rand'])
loss.backward()
R.attend = orig

vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
vb.load_state_dict(g['state'], strict=False); vb = vb.to(dev)
w = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
with rng_override(x0=g['x0'], times=g['times'], frac_lengths=g['frac'], rand=g['rand']):
    l = w(g['x1'].to(dev))
print('loss', float(l), float(loss))
eng = vb.engine(2, 40, True)
B, H, Np, D, I = 2, 2, 56, 64, 128
gflat = torch.zeros(vb._flat.numel, device=dev)
eng.m.grads = gflat.data_ptr()
rt = E._rt(); st = _lib.current_stream()
assert rt.vbx_model_backward_head(C.byref(eng.m), C.byref(eng.io), None, st) == 0
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())
for layer in (1, 0):
    assert rt.vbx_model_backward_layer(C.byref(eng.m), C.byref(eng.io), layer, st) == 0
    torch.cuda.synchronize()
    q, k, v, o = cap[layer]
    dO = eng.debug_tensor('dO', -1, (B, Np, H, 64), torch.bfloat16).permute(0, 2, 1, 3)
    print(layer, 'dO', rel(dO, o.grad))
    print(layer, 'o ', rel(eng.debug_tensor('o', layer, (B, Np, H, 64), torch.bfloat16).permute(0, 2, 1, 3), o))
    print(layer, 'q16', rel(eng.debug_tensor('q16', layer, (B, H, Np, 64), torch.float16), q))
    print(layer, 'k16', rel(eng.debug_tensor('k16', layer, (B, H, Np, 64), torch.float16), k))
    print(layer, 'v', rel(eng.debug_tensor('v', layer, (B, H, Np, 64), torch.bfloat16), v))
    delta = eng.debug_tensor('delta', -1, (B, H, Np), torch.float32)
    print(layer, 'delta', rel(delta, (o.grad * o).sum(-1)))
    print(layer, 'dq', rel(eng.debug_tensor('dq', -1, (B, H, Np, 64), torch.float32), q.grad))
    print(layer, 'dk', rel(eng.debug_tensor('dk', -1, (B, H, Np, 64), torch.float32), k.grad))
    dqkv = eng.debug_tensor('dqkv', -1, (B, Np, 3, H, 64), torch.bfloat16)
    print(layer, 'dv', rel(dqkv[:, :, 2].permute(0, 2, 1, 3), v.grad))
    lse = eng.debug_tensor('lse', layer, (B, H, Np), torch.float32)
    sim = torch.einsum('bhid,bhjd->bhij', q, k) * 10
    print(layer, 'lse', float((lse.cpu().double() - torch.logsumexp(sim, -1).detach() / 0.6931471805599453).abs().max()))
