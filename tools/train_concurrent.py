"""Experiment: forward + backward of a batch as TWO concurrent half-batches on two streams (own activation arenas and gradient
buffers, shared packed weights) against one full-batch pass -- the training analogue of the sampler's concurrent halves.
    python tools/train_concurrent.py
"""
import os, sys, time, types
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from voicebox_pytorch_amd.masks import mask_from_frac_lengths  # noqa: E402

args = types.SimpleNamespace(dim=512, depth=12, heads=16, gateloop=False)
dev = torch.device("cuda:0")
vbx, vb, wrap = bench.build_model(args, dev)
vb.train()
B, N, D = 8, 1024, 512
g = torch.Generator().manual_seed(1)
x = torch.randn(B, N, D, generator=g).to(dev); flow = torch.randn(B, N, D, generator=g).to(dev)
times = torch.rand(B, generator=g).to(dev)
cmask = mask_from_frac_lengths(N, (0.7 + 0.3 * torch.rand(B, generator=g)).to(dev))
fp = vb.flat_params()
g8 = torch.zeros_like(fp.flat); ga = torch.zeros_like(fp.flat); gb = torch.zeros_like(fp.flat)
e8 = vb.engine(B, N, True)
ea = vb.engine(B // 2, N, True)
eb = vb.engine(B // 2, N, True, slot=1, wpack_from=ea)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
half = torch.full((1,), 0.5, device=dev)


def full():
    e8.forward(x, flow, cmask, times, target=flow, loss_mask=cmask)
    e8.backward(g8)


def part(e, gbuf, sl):
    e.forward(x[sl], flow[sl], cmask[sl], times[sl], target=flow[sl], loss_mask=cmask[sl])
    e.backward(gbuf, gscale=half)


def halves(concurrent):
    if concurrent:
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            part(ea, ga, slice(0, 4))
        with torch.cuda.stream(s2):
            part(eb, gb, slice(4, 8))
        cur.wait_stream(s1); cur.wait_stream(s2)
    else:
        part(ea, ga, slice(0, 4)); part(eb, gb, slice(4, 8))


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for rep in range(2):
    print(f"B=8 forward+backward, one stream : {timed(full):7.3f} ms")
    print(f"2 x B=4 back to back             : {timed(lambda: halves(False)):7.3f} ms")
    print(f"2 x B=4 on two streams           : {timed(lambda: halves(True)):7.3f} ms")
gs = ga + gb
print("grad check: |g8| %.4e  |ga+gb - g8| / |g8| = %.3e   loss %.6f vs %.6f" % (g8.norm().item(), ((gs - g8).norm() / g8.norm()).item(),
      e8.loss.item(), 0.5 * (ea.loss.item() + eb.loss.item())))
