#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b18; mkdir -p $O
timeout 600 python -m pytest tests/test_dp_gpu.py -q -k "deferred or factor_mode" > $O/t.log 2>&1; tail -3 $O/t.log
timeout 600 python bench.py > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/r05_bench_train.json; python -c "
import json; d=json.load(open('$O/r05_bench_train.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['sample']['ms'], d['cpu_baseline']['value']); print([k['stage'] for k in d['roofline']['kernels']])"
