#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/call6; mkdir -p $O
VBX_ATTN_BWD2=3 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attn" > $O/ops.log 2>&1; tail -4 $O/ops.log
VBX_ATTN_BWD2=3 timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q > $O/model.log 2>&1; tail -4 $O/model.log
for i in 1 2; do
  echo "train base    $(tools/bv.sh)" | tee -a $O/summary.log
  echo "train bwd2=1  $(VBX_ATTN_BWD2=1 tools/bv.sh)" | tee -a $O/summary.log
  echo "train bwd2=2  $(VBX_ATTN_BWD2=2 tools/bv.sh)" | tee -a $O/summary.log
  echo "train bwd2=3  $(VBX_ATTN_BWD2=3 tools/bv.sh)" | tee -a $O/summary.log
done
