"""How well is the REFERENCE's own depth-12 loss defined at its own initialisation?  (build container only: needs /root/reference)

The north star asks for "loss within 1e-3 of reference".  At the reference's initialisation the qk-normed logits have std ~80, the
softmax is nearly one-hot and the 12-layer map is chaotic.  This tool runs the UNMODIFIED reference's CPU path on the inputs of
tests/golden/cfg4_seeds.pt with different intra-op thread counts (a different fp32 summation order inside the SAME MKL / oneDNN
kernels -- no code change at all), the fp32 restatement (same mathematics, different operation order) and the fp64 restatement
(the exact value), and prints every loss next to the golden one.  The spread between the reference's OWN runs is the floor below
which no other implementation's difference from "the reference" means anything.

    python tools/reference_noise.py [seed ...]        (default: 10 11)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle import ref_loader, restate  # noqa: E402
import make_golden  # noqa: E402


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [10, 11]
    ref = ref_loader.load_reference()
    g = torch.load(os.path.join(ROOT, "tests", "golden", "cfg4_seeds.pt"), map_location="cpu", weights_only=False)
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    for s in seeds:
        rec = g[s]
        b = rec["batch"]
        state = restate.init_state_dict(cfg, seed=s)
        x1 = torch.randn(b, 1024, 512, generator=torch.Generator().manual_seed(100 + s))
        x0, times, frac, rand = make_golden.replay_draws(x1, seed=200 + s)
        gold = float(rec["loss"])
        print(f"seed {s} (B = {b}): golden loss {gold:.7f}", flush=True)
        for th in (8, 1, 3, 5):
            torch.set_num_threads(th)
            vb, wrapper = make_golden.build_reference(ref, cfg, state=state)
            torch.manual_seed(200 + s)
            t0 = time.time()
            with torch.no_grad():
                loss = float(wrapper(x1))
            print(f"  unmodified reference, {th} thread(s): {loss:.7f}  (golden {loss - gold:+.2e})  [{time.time() - t0:.0f} s]", flush=True)
            del vb, wrapper
        torch.set_num_threads(8)
        with torch.no_grad():
            l32 = float(restate.cfm_loss(state, cfg, x1, x0, times, frac, rand))
            p64 = {k: v.double() for k, v in state.items()}
            l64 = float(restate.cfm_loss(p64, cfg, x1.double(), x0.double(), times.double(), frac, rand))
        print(f"  fp32 restatement: {l32:.7f}  (golden {l32 - gold:+.2e});  fp64 restatement (exact): {l64:.7f}  (golden {l64 - gold:+.2e})", flush=True)


if __name__ == "__main__":
    main()
