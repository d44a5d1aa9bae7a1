#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/attn_bwd_accuracy.py > $O/bwd_accuracy.txt 2>&1; grep -v amdgpu $O/bwd_accuracy.txt
VBX_ATTN_BWD_FOLD=0 timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -s -k "padded or small_golden_loss" > $O/nofold_tests.log 2>&1; grep -n "worst\|relative grad errors\|passed\|failed" $O/nofold_tests.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q -s > $O/suite.log 2>&1; grep -n "passed\|failed\|^FAILED" $O/suite.log | tail
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o run -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample > $R/$O/prof_train.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_train -name "*.db" | head -1) 9 > $O/train_step_kernel_stats.txt 2>&1
rm -rf $O/prof_train
head -40 $O/train_step_kernel_stats.txt | cut -c1-150
