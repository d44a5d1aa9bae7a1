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
print(f"total kernel time {tot/1e3:.2f} ms over {steps:g} steps = {tot/1e3/steps:.3f} ms/step")
print(f"{'kernel':92s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'%':>6s}")
for n, c, t, a in rows[:45]:
    print(f"{short(n):92s} {c:6d} {t:10.1f} {a:9.2f} {100*t/tot:6.2f}")
