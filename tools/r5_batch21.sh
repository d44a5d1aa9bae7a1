#!/bin/bash
# gradient norm from the slab-reduce partials: tests + A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/b21; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py -q -x -k "layer_reduce or splitk or gradient_norm or factor_mode or clip_adam or adamw or deferred or sumsq" > $O/t.log 2>&1; tail -4 $O/t.log
for i in 1 2; do for v in 1 0; do
VBX_SUMSQ_FOLD=$v timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sumsq_fold=$v', d['ms_per_step'], d['final_loss'])"
done; done 2>&1 | tee $O/ab.log
