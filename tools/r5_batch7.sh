#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/suite.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/suite.log | tail -20
timeout 600 python bench.py --no-cpu-baseline --no-sample > $O/bench_factors.json 2> $O/bench_factors.err
timeout 600 python bench.py --no-cpu-baseline --no-sample --adaln-exchange materialize > $O/bench_mat.json 2> $O/bench_mat.err
timeout 600 python bench.py --no-cpu-baseline --no-sample > $O/bench_factors2.json 2> $O/bench_factors2.err
python - <<PY
import json
for f in ("bench_factors","bench_mat","bench_factors2"):
    d=json.loads(open("$O/"+f+".json").read().strip().split("\n")[-1])
    print(f, d["ms_per_step"], d.get("adaln_grads"), d.get("final_loss"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o run -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample > $R/$O/prof_train.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_train -name "*.db" | head -1) 9 > $O/train_step_kernel_stats.txt 2>&1
rm -rf $O/prof_train
head -32 $O/train_step_kernel_stats.txt | cut -c1-150
