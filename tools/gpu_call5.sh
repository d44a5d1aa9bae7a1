#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/call5; mkdir -p $O
VBX_ATTN_BWD2=${BWD2:-1} timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attn" > $O/ops.log 2>&1; tail -4 $O/ops.log
VBX_ATTN_BWD2=${BWD2:-1} timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q > $O/model.log 2>&1; tail -4 $O/model.log
for i in 1 2; do
  echo "train base  $(tools/bv.sh)" | tee -a $O/summary.log
  echo "train bwd2=${BWD2:-1}  $(VBX_ATTN_BWD2=${BWD2:-1} tools/bv.sh)" | tee -a $O/summary.log
done
VBX_ATTN_BWD2=${BWD2:-1} timeout 300 python bench.py --no-cpu-baseline --no-sample > $O/bench_bwd2.json 2> $O/bench_bwd2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/call5/bench_bwd2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"]); [print(k["stage"], k["us_per_launch"], k.get("frac")) for k in d["roofline"]["kernels"][:6]]
PY
