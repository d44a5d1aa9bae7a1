"""Run-to-run pattern of the 64-interval sampler: ten back-to-back runs and six runs separated by 0.5 s pauses (python tools/sample_runs.py)."""
import os, sys, time, types, torch
sys.path.insert(0, os.getcwd())
import bench
args = types.SimpleNamespace(dim=512, depth=12, heads=16, gateloop=False, attn_dropout=0.0, ff_dropout=0.0)
dev = torch.device("cuda:0")
vbx, vb, wrap = bench.build_model(args, dev)
vb.eval()
x = torch.randn(8, 1024, 512, device=dev)
with torch.no_grad():
    wrap.sample(cond=x, steps=65); torch.cuda.synchronize()
    ts = []
    for i in range(10):
        t0 = time.perf_counter(); wrap.sample(cond=x, steps=65); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("back to back   :", " ".join(f"{t:.1f}" for t in ts))
    ts = []
    for i in range(6):
        time.sleep(0.5)
        t0 = time.perf_counter(); wrap.sample(cond=x, steps=65); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("0.5 s pauses   :", " ".join(f"{t:.1f}" for t in ts))
