# Round-2 measurement bundle (GPU box, through gpurun): kernel-trace stats of the train step and of a sample, PMC passes
# (HBM traffic and MFMA busy) of the train step -> gpurun_out/r02/; summaries are copied into profiles/ afterwards.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_train -o run -- $B > $O/prof_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sample -o run -- python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 8 --no-cpu-baseline > $O/prof_sample.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o run -- $B > $O/pmc_mfma.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_train -name "*.db" | head -1) 9 > $O/r02_train_step_kernel_stats.txt 2>&1
python tools/prof_summary.py $(find $O/prof_sample -name "*.db" | head -1) 18 > $O/r02_sample_kernel_stats.txt 2>&1
python tools/pmc_summary.py $O/r02_train_pmc.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma > $O/r02_train_pmc.txt 2>&1
# the raw traces are large: keep only the summaries
rm -rf $O/prof_train $O/prof_sample $O/pmc_fetch $O/pmc_write $O/pmc_mfma
head -30 $O/r02_train_step_kernel_stats.txt
