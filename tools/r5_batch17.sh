#!/bin/bash
# weight gradients on the side stream (round-2 mode) re-measured on the round-5 kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/b17; mkdir -p $O
for i in 1 2; do
for v in 0 1; do
VBX_WGRAD_STREAM=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_stream=$v', d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
