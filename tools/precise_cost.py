"""Cost of the exact-operand ("precise") mode next to the fast path at the benchmark shape (GPU box).

    python tools/precise_cost.py [--dim 512] [--batch 8]

Times, with HIP events on the current stream: the training objective's forward + backward (ConditionalFlowMatcherWrapper.forward +
loss.backward) and one inference forward (VoiceBox.forward in eval mode), in both modes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import voicebox_pytorch_amd as vbx  # noqa: E402


def timed(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vb = vbx.VoiceBox(dim=a.dim, depth=12, dim_head=64, heads=16, condition_on_text=False).to(dev)
    w = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    x = torch.randn(a.batch, 1024, a.dim, device=dev)
    t = torch.full((a.batch,), 0.37, device=dev)

    def train():
        vb.train()
        vb.zero_grad(set_to_none=True)
        w(x).backward()

    def infer():
        vb.eval()
        with torch.no_grad():
            vb(x, times=t, cond_token_ids=None, cond=x, cond_drop_prob=0.0)

    out = {}
    for mode in ("fast", "precise"):
        vbx.set_precise(mode == "precise")
        out[mode] = (timed(train), timed(infer))
    vbx.set_precise(False)
    print(f"dim {a.dim}, depth 12, {a.batch} x 1024 frames: forward + backward / inference forward, ms")
    for mode, (tr, inf) in out.items():
        print(f"  {mode:8s} {tr:8.2f} {inf:8.2f}")
    print(f"  ratio    {out['precise'][0] / out['fast'][0]:8.2f} {out['precise'][1] / out['fast'][1]:8.2f}")


if __name__ == "__main__":
    main()
