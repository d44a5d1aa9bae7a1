#!/bin/bash
# Round 3: kernel stats and HBM-traffic PMC passes of the train step with the ONE-PASS attention backward selected
# (VBX_ATTN_BWD_ONEPASS=1), and kernel stats of the step with dropout 0.1 / 0.1 -> gpurun_out/op/r03_onepass_* , r03_dropout_*
cd $GRAFT_REPO_ROOT; O=gpurun_out/op; mkdir -p $O
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample"
VBX_ATTN_BWD_ONEPASS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_op -o run -- $B > $R/$O/prof_op.log 2>&1
VBX_ATTN_BWD_ONEPASS=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o run -- $B > $R/$O/pmc_fetch.log 2>&1
VBX_ATTN_BWD_ONEPASS=1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o run -- $B > $R/$O/pmc_write.log 2>&1
VBX_ATTN_BWD_ONEPASS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_mfma -o run -- $B > $R/$O/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_drop -o run -- $B --attn-dropout 0.1 --ff-dropout 0.1 > $R/$O/prof_drop.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_op -name "*.db" | head -1) 9 > $O/r03_onepass_train_step_kernel_stats.txt 2>&1
python tools/prof_summary.py $(find $O/prof_drop -name "*.db" | head -1) 9 > $O/r03_dropout_train_step_kernel_stats.txt 2>&1
python tools/pmc_summary.py $O/r03_onepass_train_pmc.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma > $O/r03_onepass_train_pmc.txt 2>&1
rm -rf $O/prof_op $O/prof_drop $O/pmc_fetch $O/pmc_write $O/pmc_mfma
head -8 $O/r03_onepass_train_step_kernel_stats.txt; grep -h "attn_bwd1\|attn_delta" $O/r03_onepass_train_pmc.txt | cut -c1-400; head -12 $O/r03_dropout_train_step_kernel_stats.txt
