"""Summarise a rocprofv3 --kernel-trace (--stats) sqlite .db or kernel_trace.csv into a per-kernel table."""
import sys, sqlite3, csv, collections, re

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"void ", "", n)
    return n.split("(")[0][:90]

def from_db(path):
    con = sqlite3.connect(path)
    return [(r[0], r[1], r[2], r[3]) for r in con.execute("select name,total_calls,total_duration,average from top_kernels")]

def from_csv(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += d
    return [(k, v[0], v[1], v[1] / v[0]) for k, v in agg.items()]

rows = from_db(sys.argv[1]) if sys.argv[1].endswith(".db") else from_csv(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
# one-time setup, not part of a step: parameter flattening (one device copy per nn.Parameter) and the first weight packing
isteps = int(steps)
SETUP = re.compile(r"copyBuffer|pack_weight_kernel|pack_bias_kernel")  # (their call counts are not multiples of the step count)
per_step = [r for r in rows if not (SETUP.search(r[0]) and isteps > 0 and r[1] % isteps != 0)]
setup = [r for r in rows if r not in per_step]
mfma = lambda n: re.search(r"gemm|attn_(fwd|bwd)", n) is not None
print(f"total kernel time {tot/1e3:.2f} ms over {steps:g} steps = {tot/1e3/steps:.3f} ms/step")
if isteps > 1:
    ps_t, ps_c = sum(r[2] for r in per_step), sum(r[1] for r in per_step)
    mf_t, mf_c = sum(r[2] for r in per_step if mfma(r[0])), sum(r[1] for r in per_step if mfma(r[0]))
    print(f"per-step kernels: {ps_c/steps:.0f} launches, {ps_t/1e3/steps:.3f} ms per step (MFMA kernels {mf_c/steps:.0f} launches, "
          f"{mf_t/1e3/steps:.3f} ms; others {(ps_c-mf_c)/steps:.0f} launches, {(ps_t-mf_t)/1e3/steps:.3f} ms); one-time setup: "
          f"{sum(r[1] for r in setup)} launches, {sum(r[2] for r in setup)/1e3:.2f} ms in total")
print(f"{'kernel':92s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'%':>6s}")
for n, c, t, a in rows[:45]:
    print(f"{short(n):92s} {c:6d} {t:10.1f} {a:9.2f} {100*t/tot:6.2f}" + ("" if (n, c, t, a) in per_step or isteps <= 1 else "  (setup)"))
