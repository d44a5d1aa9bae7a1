#!/bin/bash
# Round-end verification + measurement bundle (GPU box): full GPU test suite, smoke, the default bench line (train + sample leg +
# CPU baseline), the sample-mode and dim-1024 bench lines, then the rocprofv3 bundle (kernel stats of train step and sample, three
# PMC passes) -> gpurun_out/<tag>/<tag>_*; the summaries are copied into profiles/ afterwards.
#   usage: gpurun --timeout 2400 -- 'bash tools/round_bundle.sh r05'
T=${1:-r05}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$T; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/${T}_bench_train.json; tail -c 700 $O/${T}_bench_train.json
timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${T}_bench_sample.json
timeout 300 python bench.py --dim 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 > $O/${T}_bench_train_dim1024.json
python - <<PY
import json
for n in ("${T}_bench_train","${T}_bench_sample","${T}_bench_train_dim1024"):
    try:
        d=json.loads(open(f"$O/{n}.json").read())
        print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), d.get("sample",{}).get("ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sample"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_train -o run -- $B > $R/$O/prof_train.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_sample -o run -- python $R/bench.py --mode sample --steps 1 --warmup 0 --intervals 8 --no-cpu-baseline > $R/$O/prof_sample.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o run -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o run -- $B > $R/$O/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_mfma -o run -- $B > $R/$O/pmc_mfma.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_train -name "*.db" | head -1) 9 > $O/${T}_train_step_kernel_stats.txt 2>&1
python tools/prof_summary.py $(find $O/prof_sample -name "*.db" | head -1) 18 > $O/${T}_sample_kernel_stats.txt 2>&1
python tools/pmc_summary.py $O/${T}_train_pmc.json $O/pmc_fetch $O/pmc_write $O/pmc_mfma > $O/${T}_train_pmc.txt 2>&1
rm -rf $O/prof_train $O/prof_sample $O/pmc_fetch $O/pmc_write $O/pmc_mfma
head -24 $O/${T}_train_step_kernel_stats.txt
bash tools/precise_report.sh $T > /dev/null 2>&1; tail -12 $O/${T}_precise_parity.txt
