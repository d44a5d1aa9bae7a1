#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -s > $O/suite.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/suite.log | tail
grep -n "reference init\|precise mode mean" $O/suite.log | cut -c1-330
for i in 1 2; do timeout 300 python tools/attn_bench.py 50 >> $O/attn_bench.txt 2>&1; done; grep -v amdgpu $O/attn_bench.txt
timeout 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err
python - <<PY
import json
d=json.loads(open("$O/bench_train.json").read().strip().split("\n")[-1])
print("train ms", d["ms_per_step"], "sample", d.get("sample"))
PY
