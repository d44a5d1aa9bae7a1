#!/bin/bash
# Round 3: whole GPU suite with the printed parity numbers of the new golden tests, smoke, short benches
cd $GRAFT_REPO_ROOT; O=gpurun_out/c3; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rA > $O/pytest.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
grep -h "cfg4 reference-init\|cfg3 \|cfg5 B=8\|relative grad errors vs REFERENCE\|grad-norm rel errors vs REFERENCE\|cfg1 loss\|cfg4 (depth 12) loss\|cfg4_wc (depth" $O/pytest.log | cut -c1-900 | sort -u
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/r03_mid_bench_train.json; python - <<'PY'
import json
d=json.loads(open("gpurun_out/c3/r03_mid_bench_train.json").read())
print("train", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], "sample", d.get("sample",{}).get("ms"), d.get("sample",{}).get("fwd_frac"))
print("cpu", d.get("cpu_baseline"))
PY
