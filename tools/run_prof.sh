set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu --timeout 900 -p no:cacheprovider 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_train.json
python bench.py --mode sample --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_sample.json
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --gateloop 2>&1 | tail -1 > gpurun_out/bench_train_gateloop.json
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof6 -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof6.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof6gl -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --gateloop > $R/gpurun_out/prof6gl.log 2>&1
cd $R
cat gpurun_out/bench_train.json gpurun_out/bench_sample.json gpurun_out/bench_train_gateloop.json
ls gpurun_out/prof6 gpurun_out/pmc_fetch
