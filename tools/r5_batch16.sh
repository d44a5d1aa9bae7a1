#!/bin/bash
# grouped weight-gradient launch: K-split sweep in situ (198 workgroups at 3 splits leave 58 CUs idle)
cd $GRAFT_REPO_ROOT; O=gpurun_out/b16; mkdir -p $O
for s in 3 4 5 6 7 8 2 3; do
VBX_WGRAD_SPLITS3=$s timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['stage']:x['us_per_launch'] for x in d['roofline']['kernels']}
print('splits3=$s', d['ms_per_step'], 'wgrad', k.get('wgrad (4 GEMMs)'), 'slab reduce', k.get('wgrad slab reduce'))"
done 2>&1 | tee $O/sweep.log
