# Round-end check on the GPU box (one gpurun call): full GPU suite, smoke, benches, kernel-trace profile of the train step.
# Every step is bounded by its own timeout; logs land in gpurun_out/.
set -x
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout 150 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log
timeout 60 python __graft_entry__.py --smoke > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
timeout 150 python bench.py > gpurun_out/bench_train_full.log 2>&1; tail -1 gpurun_out/bench_train_full.log > gpurun_out/bench_train.json
timeout 60 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_sample.json
export TMPDIR=/tmp
cd /tmp
timeout 90 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof11 -o run -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof11.log 2>&1
cd $R
cat gpurun_out/bench_train.json gpurun_out/bench_sample.json
