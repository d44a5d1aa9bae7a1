"""Experiment: the two half-batch integrations of the sampler as two INDEPENDENT graphs on two streams (no join between intervals),
the second one started with a delay, so that different kernels of the two forwards overlap (attention beside GEMMs) instead of the
same ones.    python tools/sample_offset.py [intervals]"""
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
with torch.no_grad():
    smp = MidpointSampler(vb, B, N, steps, split=2)
    ref = smp.run(y0, cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ref = smp.run(y0, cond); torch.cuda.synchronize()
    print(f"one graph, two joined branches : {(time.perf_counter() - t0) * 1e3:8.2f} ms")
    # separate graphs per part
    graphs = []
    for p in smp.parts:
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            smp._interval_part(p)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            smp._interval_part(p)
        graphs.append(gr)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for delay_us in (0, 30, 60, 100, 150, 250, 400):
        best = 1e9
        for rep in range(3):
            smp.y.copy_(y0); smp.cond.copy_(cond); smp.counters.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(streams[1]):
                if delay_us:
                    torch.cuda._sleep(int(delay_us * 2100))
            for _ in range(iv):
                for h in range(2):
                    with torch.cuda.stream(streams[h]):
                        graphs[h].replay()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        print(f"two graphs, second delayed {delay_us:4d} us: {best:8.2f} ms   max |diff| {(smp.y - ref).abs().max().item():.1e}")
