"""Experiment: ODE sampling of a batch as TWO concurrent half-batch hipGraphs on two streams.
Every kernel of a forward has a ramp, a drain and (GEMMs) a VALU-bound epilogue during which the matrix pipes idle
(tools/native/gemm_trace.cpp); a second, independent stream of kernels can fill those holes.  Batch elements are independent in
every kernel of the path, so the halves must reproduce the full-batch result.
    python tools/sample_concurrent.py [intervals]
"""
import os, sys, time, types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from voicebox_pytorch_amd.solver import MidpointSampler  # noqa: E402

iv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
args = types.SimpleNamespace(dim=512, depth=12, heads=16, gateloop=False)
dev = torch.device("cuda:0")
vbx, vb, wrap = bench.build_model(args, dev)
vb.eval()
B, N, D = 8, 1024, 512
g = torch.Generator().manual_seed(1)
cond = torch.randn(B, N, D, generator=g).to(dev)
y0 = torch.randn(B, N, D, generator=g).to(dev)
steps = iv + 1


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


with torch.no_grad():
    full = MidpointSampler(vb, B, N, steps, split=1)
    ms_full, out_full = timed(lambda: full.run(y0, cond))
    print(f"B=8, one stream                  : {ms_full:8.2f} ms  ({ms_full / (2 * iv):.3f} ms per NFE)")
    for split in (2, 4):
        smp = MidpointSampler(vb, B, N, steps, split=split)
        ms, out = timed(lambda: smp.run(y0, cond))
        print(f"B=8 as {split} concurrent parts, 1 graph: {ms:8.2f} ms  ({ms / (2 * iv):.3f} ms per NFE)   max |diff| vs one stream "
              f"{(out - out_full).abs().max().item():.3e}")
