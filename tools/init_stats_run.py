"""Fast-path loss minus reference over the seeds of tests/golden/init_stats.pt (cfg4: 16 seeds; cfg3: 6) and 12 dim-64 seeds, for
whatever tree this file is run in (python tools/init_stats_run.py from the tree's root): the A/B tool behind the parity statistics."""
import os, sys, torch
ROOT = os.getcwd()
sys.path.insert(0, ROOT)
import voicebox_pytorch_amd as vbx
from voicebox_pytorch_amd.masks import rng_override
from oracle import restate
dev = "cuda"
def build(dim, depth, heads, state):
    vb = vbx.VoiceBox(dim=dim, num_cond_tokens=500, depth=depth, dim_head=64, heads=heads, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    vb = vb.to(dev)
    return vb, vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
def stats(ds):
    n = len(ds); return sum(abs(d) for d in ds) / n, (sum(d * d for d in ds) / n) ** 0.5, max(abs(d) for d in ds)
g = torch.load(os.path.join(ROOT, "tests/golden/init_stats.pt"), map_location="cpu", weights_only=False)
which = sys.argv[1:] or ["cfg4", "small"]
for tag, dim in (("cfg4", 512), ("cfg3", 1024)):
    if tag not in which: continue
    cfg = restate.Cfg(dim=dim, depth=12, heads=16, dim_head=64)
    ds = []
    for (t, s_), rec in sorted(g.items()):
        if t != tag: continue
        state = restate.init_state_dict(cfg, seed=s_)
        vb, wrapper = build(dim, 12, 16, state)
        x1 = torch.randn(2, 1024, dim, generator=torch.Generator().manual_seed(100 + s_))
        torch.manual_seed(200 + s_)
        x0 = torch.randn_like(x1)
        with torch.no_grad(), rng_override(x0=x0, times=rec["times"], frac_lengths=rec["frac"], rand=rec["rand"]):
            ds.append(float(wrapper(x1.to(dev))) - rec["loss"])
        del vb, wrapper, state; torch.cuda.empty_cache()
    print(tag, "mean|d| %.2e rms %.2e max %.2e" % stats(ds), [round(d, 5) for d in ds], flush=True)
if "small" in which:
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    ds = []
    for s_ in range(48):
        state = restate.init_state_dict(cfg, seed=50 + s_)
        vb, wrapper = build(64, 2, 2, state)
        gen = torch.Generator().manual_seed(150 + s_)
        x1, x0 = torch.randn(2, 96, 64, generator=gen), torch.randn(2, 96, 64, generator=gen)
        times, frac, rand = torch.rand(2, generator=gen), 0.7 + 0.3 * torch.rand(2, generator=gen), torch.rand(2, generator=gen)
        with torch.no_grad():
            ref = float(restate.cfm_loss(state, cfg, x1, x0, times, frac, rand))
            with rng_override(x0=x0, times=times, frac_lengths=frac, rand=rand):
                ds.append(float(wrapper(x1.to(dev))) - ref)
    print("small(48 seeds) mean|d| %.2e rms %.2e max %.2e" % stats(ds), [round(d, 5) for d in ds[:12]], flush=True)
