#!/bin/bash
# Round 3: dropout op / model tests first (fast feedback), then the whole GPU suite, then the dropout A/B of the train step
cd $GRAFT_REPO_ROOT; O=gpurun_out/c4; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "dropout or philox" -rA > $O/pytest_dropout.log 2>&1; echo "dropout rc=$?"; grep -E "passed|failed" $O/pytest_dropout.log | tail -2; grep -E "^FAILED|^ERROR|^E  " $O/pytest_dropout.log | head -30; grep -h "dropout step" $O/pytest_dropout.log | head -3
timeout 1200 python -m pytest tests -m gpu -q -rA > $O/pytest.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head
for a in "" "--attn-dropout 0.1 --ff-dropout 0.1" "--ff-dropout 0.1" "--attn-dropout 0.1"; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample $a 2>$O/bench_err.log | tail -1 > $O/b.json
  python - "$a" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/c4/b.json").read())
    print("bench", sys.argv[1] or "(no dropout)", d["ms_per_step"], "ms", [(k["stage"], k["us_per_launch"]) for k in d["roofline"]["kernels"] if "attention" in k["stage"]])
except Exception as e:
    print("bench", sys.argv[1], "ERR", e, open("gpurun_out/c4/bench_err.log").read()[-600:])
PY
done
