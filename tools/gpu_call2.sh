#!/bin/bash
# GPU call 2: gemm4 validation + three-way GEMM timing, in-situ stage tables for the old and the new dispatch, parity tests on both
cd $GRAFT_REPO_ROOT; O=gpurun_out/call2; mkdir -p $O
for part in correct race time; do
  timeout 200 tools/native/gemm3_check $part > $O/g3_$part.log 2>&1; echo "gemm3_check $part rc=$?" | tee -a $O/summary.log
  tail -3 $O/g3_$part.log
done
VBX_GEMM_PATH=1 timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "cfg4 or well_conditioned or packed_weights or flash" > $O/pytest_new_path1.log 2>&1; tail -3 $O/pytest_new_path1.log
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_auto.log 2>&1; tail -8 $O/pytest_auto.log
VBX_GEMM_PATH=1 timeout 300 python bench.py --no-cpu-baseline > $O/bench_path1.json 2> $O/bench_path1.err; tail -c 300 $O/bench_path1.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_auto.json 2> $O/bench_auto.err; tail -c 300 $O/bench_auto.json
VBX_GEMM_PATH=2 timeout 300 python bench.py --no-cpu-baseline --no-sample > $O/bench_path2.json 2> $O/bench_path2.err
for i in 1 2; do
  echo "train path1 $(VBX_GEMM_PATH=1 tools/bv.sh)" | tee -a $O/summary.log
  echo "train auto  $(tools/bv.sh)" | tee -a $O/summary.log
done
