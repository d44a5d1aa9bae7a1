#!/bin/bash
# GPU call 1 of round 2: gemm3 standalone check, full GPU test suite on the default (gemm3 + grouped wgrad) path, bench A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/call1; mkdir -p $O
for part in correct race time; do
  timeout 150 tools/native/gemm3_check $part > $O/g3_$part.log 2>&1; echo "gemm3_check $part rc=$?" | tee -a $O/summary.log
  tail -4 $O/g3_$part.log
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest_default.log; tail -5 $O/pytest_default.log
for i in 1 2; do
  echo "train old   $(VBX_GEMM3=0 tools/bv.sh)" | tee -a $O/summary.log
  echo "train new   $(tools/bv.sh)" | tee -a $O/summary.log
  echo "train new nogroup $(VBX_GROUP_WGRAD=0 tools/bv.sh)" | tee -a $O/summary.log
done
echo "sample old  $(VBX_GEMM3=0 tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
echo "sample new  $(tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
