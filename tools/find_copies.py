"""Where do the per-step device copies (__amd_rocclr_copyBuffer) of the train step come from?  torch.profiler with Python stacks.
usage (GPU box): python tools/find_copies.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    dim, depth, heads, gateloop, batch, frames = 512, 12, 16, False, 8, 1024


dev = torch.device("cuda", 0)
vbx, vb, wrapper = bench.build_model(A, dev)
from voicebox_pytorch_amd.dp import TrainStep  # noqa: E402

ts = TrainStep(wrapper, lr=3e-4, max_grad_norm=0.5)
x = torch.randn(A.batch, A.frames, A.dim, device=dev)
for _ in range(3):
    ts.step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        ts.step(x)
    torch.cuda.synchronize()
names = collections.Counter()
for e in prof.events():
    n = e.name
    if "Memcpy" in n or "memcpy" in n or "copyBuffer" in n or "copy_" in n or "fillBuffer" in n or "Memset" in n:
        stack = [s for s in (e.stack or []) if "voicebox" in s or "bench" in s or "dp.py" in s]
        names[(n[:60], stack[0][:110] if stack else "")] += 1
for (n, st), c in names.most_common(40):
    print(f"{c:5d}  {n:60s}  {st}")
