#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b19; mkdir -p $O
for i in 1 2; do for s in 2 4; do
VBX_SAMPLE_SPLIT=$s timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('split=$s', d['ms_per_step'], d['value'])"
done; done 2>&1 | tee $O/ab.log
