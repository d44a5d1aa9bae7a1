#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
(cd .oldtree && timeout 600 python tools/init_stats_run.py cfg4 small > $O/stats_old.txt 2>&1)
timeout 600 python tools/init_stats_run.py cfg4 small > $O/stats_new.txt 2>&1
grep -v amdgpu.ids $O/stats_old.txt; grep -v amdgpu.ids $O/stats_new.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "attn or delta or rmsnorm or qkv or golden or cfg1 or padded" > $O/suite_part.log 2>&1; tail -5 $O/suite_part.log
timeout 300 python bench.py --no-cpu-baseline --no-sample > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
VBX_DELTA_FUSED=0 timeout 300 python bench.py --no-cpu-baseline --no-sample > $O/bench_nodelta.json 2> $O/bench_nodelta.err; tail -c 300 $O/bench_nodelta.json
