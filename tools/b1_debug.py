"""Debug helper: one fused attention backward at a given shape with both variants; where do they differ / where are non-finite values."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from voicebox_pytorch_amd import _lib as L
import test_ops_gpu as T

Bsz, H, Np = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L.lib()
c = T._bwd_case(L, Bsz, H, Np, seed=Np + Bsz)
d1, g1, _ = T._bwd_fused(L, c, 1)
I = H * 64
for r in range(reps):
    d2, g2, scratch = T._bwd_fused(L, c, 2)
    w = scratch[:64].view(torch.int32).cpu()
    bad = ~torch.isfinite(d2.float())
    print(f"rep {r}: sync words {w[:9].tolist()}  non-finite: {int(bad.sum())}", end="")
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(f"  rows {rows[:8].tolist()}..{rows[-1].item()} ({len(rows)}), cols {cols[:4].tolist()}..{cols[-1].item()} ({len(cols)}; q<{I}, k<{2*I})", end="")
    else:
        print("  q rel", T.rel_err(d2[:, :I].float(), d1[:, :I].float()), "k eq", torch.equal(d1[:, I:2*I], d2[:, I:2*I]), "v eq", torch.equal(d1[:, 2*I:], d2[:, 2*I:]), end="")
    print()
