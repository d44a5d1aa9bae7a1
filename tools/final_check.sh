#!/bin/bash
# Round-end verification bundle (GPU box): full GPU test suite, smoke, the default bench line (train + sample leg + CPU baseline),
# the sample-mode and dim-1024 bench lines.   usage: gpurun --timeout 1800 -- 'bash tools/final_check.sh'
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/r02_bench_train.json; tail -c 600 $O/r02_bench_train.json
timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_sample.json
timeout 300 python bench.py --dim 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 > $O/r02_bench_train_dim1024.json
python - <<'PY'
import json
for n in ("r02_bench_train","r02_bench_sample","r02_bench_train_dim1024"):
    try:
        d=json.loads(open(f"gpurun_out/final/{n}.json").read())
        print(n, d["value"], d["ms_per_step"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), d.get("sample",{}).get("ms"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "ERR", e)
PY
