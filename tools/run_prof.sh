# Round-end measurement bundle (run on the GPU box through gpurun): benches, kernel-trace profiles, PMC HBM-traffic passes.
set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/bench_train_full.log 2>&1; tail -1 gpurun_out/bench_train_full.log > gpurun_out/bench_train.json
python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_sample.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dim 1024 2>&1 | tail -1 > gpurun_out/bench_train_dim1024.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof10 -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof10.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof10s -o run -- python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 8 --no-cpu-baseline > $R/gpurun_out/prof10s.log 2>&1
# HBM traffic of the roofline kernel: separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), sample mode so that
# every FeedForward-in GEMM launch has the same (inference) epilogue as the roofline launches
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o run -- python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 2 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o run -- python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 2 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
cat gpurun_out/bench_train.json gpurun_out/bench_sample.json gpurun_out/bench_train_dim1024.json
