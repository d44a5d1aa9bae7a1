"""Which host-side call launches the torch fill / elementwise kernels inside a train step?  (three FillFunctor<float>
launches per step.)  Runs 2 steps under torch.profiler with stacks and prints every aten op that launches a device kernel, with the
innermost repository frame.  Usage (GPU box): python tools/find_fills.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import argparse
from torch.profiler import profile, ProfilerActivity

args = argparse.Namespace(dim=512, depth=12, heads=16, batch=8, frames=1024, gateloop=False, attn_dropout=0.0, ff_dropout=0.0)
dev = torch.device("cuda", 0)
vbx, vb, wrapper = bench.build_model(args, dev)
from voicebox_pytorch_amd.dp import TrainStep
ts = TrainStep(wrapper, lr=3e-4, max_grad_norm=0.5)
x = torch.randn(8, 1024, 512, device=dev)
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        ts.step(x)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type.name == "CPU" and ev.name.startswith("aten::") and ev.self_device_time_total > 0:
        frame = next((s for s in ev.stack if "/repo/" in s or "voicebox" in s), ev.stack[0] if ev.stack else "?")
        rows.append((ev.name, str(ev.input_shapes), ev.self_device_time_total, frame))
agg = {}
for n, sh, t, fr in rows:
    k = (n, sh, fr)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += t
for (n, sh, fr), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 2:9.1f} us/step  x{c / 2:<4.1f} {n:28s} {sh:40s} {fr}")
