#!/bin/bash
# all-layer adaLN d(time_emb) + fused layer reduce: op tests, model tests, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/b12; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "adaln or layer_reduce or multi_reduce or splitk" > $O/ops.log 2>&1; tail -4 $O/ops.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -q -x -k "small_golden or train_step or adaln or trajectory or gateloop or text or transformer" > $O/model.log 2>&1; tail -4 $O/model.log
for i in 1 2; do
for v in "1 1" "0 0" "1 0" "0 1"; do set -- $v
VBX_ADALN_BWD_ALL=$1 VBX_LAYER_REDUCE=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ada_all=$1 layer_reduce=$2', d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
