#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/call3; mkdir -p $O
for part in correct race time; do
  timeout 240 tools/native/gemm3_check $part > $O/g3_$part.log 2>&1; echo "gemm3_check $part rc=$?" | tee -a $O/summary.log
  tail -3 $O/g3_$part.log
done
for i in 1 2; do
  echo "train auto(path1 gemms + grouped gemm3 wgrad)  $(tools/bv.sh)" | tee -a $O/summary.log
  echo "train path3  $(VBX_GEMM_PATH=3 tools/bv.sh)" | tee -a $O/summary.log
done
