#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/call7; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/model.log 2>&1; tail -3 $O/model.log
for i in 1 2; do
  echo "sample ffin4=0  $(VBX_GEMM4_FFIN=0 tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
  echo "sample ffin4=1  $(tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
done
echo "train              $(tools/bv.sh)" | tee -a $O/summary.log
# the RCCL path on this one GPU: torchrun with one rank, collectives forced (VBX_FORCE_DIST=1)
export VBX_FORCE_DIST=1
for i in 1 2; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sample 2> $O/dist$i.err | tail -1 > $O/bench_dist$i.json
  python -c "import json;d=json.loads(open('$O/bench_dist$i.json').read());print('train nccl world-1 forced', d['value'], d['ms_per_step'])" | tee -a $O/summary.log
done
unset VBX_FORCE_DIST
echo "train              $(tools/bv.sh)" | tee -a $O/summary.log
