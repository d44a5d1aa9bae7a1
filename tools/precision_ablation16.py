"""Which operand class carries the factor 2 between the fast path and the fp32 floor at the reference's own initialisation?
  tools/precision_ablation.py answers on ONE seed -- a draw, not a measurement.  This runs the same emulation
(oracle/restate.py: `emulate_fp16_operands`, one operand class rounded to fp16 at a time) over the 16 config-4 seeds of
tests/golden/init_stats.pt (dim 512, depth 12, B = 2 x 1024 frames; reference losses from the unmodified reference) and reports, per
class, mean |d|, RMS and max of (loss - reference).  CPU only (~40 min on 8 cores).  Output: profiles/r06_precision_ablation.txt.
  python tools/precision_ablation16.py [nseeds]"""
import os, sys, time, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restate  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", 6)))
g = torch.load(os.path.join(ROOT, "tests", "golden", "init_stats.pt"), map_location="cpu", weights_only=False)
seeds = sorted(s for (t, s) in g if t == "cfg4")[: int(sys.argv[1]) if len(sys.argv) > 1 else 16]
cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
classes = [("fp32 restatement (no rounding)", "none", ()), ("every operand fp16", None, ()), ("only qkv (to_qkv operands)", {"qkv"}, ()),
           ("only qk (q-hat, k-hat)", {"qk"}, ()), ("only p, v", {"p", "v"}, ()), ("only out (to_out operands)", {"out"}, ()),
           ("only ff", {"ff"}, ()), ("only ada", {"ada"}, ()), ("only emb", {"emb"}, ()),
           ("all fp16, qkv as hi+lo", None, ("qkv",)), ("all fp16, qkv + qk as hi+lo", None, ("qkv", "qk")),
           ("all fp16, p + v as hi+lo", None, ("p", "v"))]
res = {name: [] for name, _, _ in classes}
t0 = time.time()
for s_ in seeds:
    rec = g[("cfg4", s_)]
    state = restate.init_state_dict(cfg, seed=s_)
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(100 + s_))
    torch.manual_seed(200 + s_)
    x0 = torch.randn_like(x1)
    for name, only, split in classes:
        with torch.no_grad():
            if only == "none":
                v = float(restate.cfm_loss(state, cfg, x1, x0, rec["times"], rec["frac"], rec["rand"]))
            else:
                restate._EMULATE["only"], restate._EMULATE["split"] = only, set(split)
                try:
                    with restate.emulate_fp16_operands():
                        v = float(restate.cfm_loss(state, cfg, x1, x0, rec["times"], rec["frac"], rec["rand"]))
                finally:
                    restate._EMULATE["only"], restate._EMULATE["split"] = None, ()
        res[name].append(v - float(rec["loss"]))
    print(f"seed {s_} done ({time.time() - t0:.0f} s): " + "  ".join(f"{res[n][-1]:+.2e}" for n, _, _ in classes), flush=True)
print(f"\nloss - reference over {len(seeds)} config-4 seeds (reference = the unmodified reference's fp32 CPU loss), x 1e-3:")
print(f"{'operand class rounded to fp16':42s} {'mean|d|':>8s} {'RMS':>8s} {'max|d|':>8s}")
for name, _, _ in classes:
    d = res[name]
    n = len(d)
    print(f"{name:42s} {sum(abs(x) for x in d) / n * 1e3:8.2f} {(sum(x * x for x in d) / n) ** 0.5 * 1e3:8.2f} {max(abs(x) for x in d) * 1e3:8.2f}")
