#!/bin/bash
# Additional round-end evidence (GPU box): PMC passes of the 64-interval sampler's forward (HBM traffic, MFMA busy) and the kernel
# statistics of the dim-1024 train step (BASELINE config 3) -> gpurun_out/extra/; summaries are copied into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; O=gpurun_out/extra; mkdir -p $R/$O; export TMPDIR=/tmp; cd /tmp
S="python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 4 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o run -- $S > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o run -- $S > $R/$O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_mfma -o run -- $S > $R/$O/pmc_mfma.log 2>&1
D="python $R/bench.py --dim 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-sample"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_d1024 -o run -- $D > $R/$O/prof_d1024.log 2>&1
cd $R
python tools/pmc_summary.py $O/r02_sample_pmc.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma > $O/r02_sample_pmc.txt 2>&1
python tools/prof_summary.py $(find $O/prof_d1024 -name "*.db" | head -1) 9 > $O/r02_train_dim1024_kernel_stats.txt 2>&1
rm -rf $O/prof_d1024 $O/pmc_fetch $O/pmc_write $O/pmc_mfma
head -8 $O/r02_sample_pmc.txt | cut -c1-200; head -14 $O/r02_train_dim1024_kernel_stats.txt
