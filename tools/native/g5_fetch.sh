#!/bin/bash
# HBM-side fetch of the gemm5 launches of `gemm5_check time` (rocprofv3 --pmc FETCH_SIZE, x 2 as MI355X_MICROARCH.md prescribes), per kernel
# instantiation, for the work maps VBX_GEMM5_PX = default / 2 / plain (VBX_GEMM5_XCD=0).     usage (GPU box): bash tools/native/g5_fetch.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export TMPDIR=/tmp; cd /tmp
for cfg in "default" "VBX_GEMM5_PX=2" "VBX_GEMM5_XCD=0"; do
  rm -rf /tmp/pf; env $( [ "$cfg" = default ] || echo $cfg ) timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o run -- $R/tools/native/gemm5_check time > /dev/null 2>&1
  python3 - "$cfg" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pf/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm5" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 200 * 256:
        k = r["Kernel_Name"]; k = k[k.index("Epi5"):k.index("Epi5") + 24]
        acc[k].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{sys.argv[1]:18s} {k:26s} launches {len(v):4d}  fetch {sum(v) / len(v) * 1024 * 2 / 1e6:7.1f} MB per launch (batch-8 and batch-4 launches mixed)")
PY
done
