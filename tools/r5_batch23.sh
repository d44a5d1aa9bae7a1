#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b23; mkdir -p $O
timeout 150 python bench.py > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/r05_bench_train.json; python -c "
import json; d=json.load(open('$O/r05_bench_train.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['sample']['ms'], d['cpu_baseline']['value'])"
