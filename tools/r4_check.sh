#!/bin/bash
# round-4 GPU check: the whole GPU suite (no -x: every failure is listed), smoke, then the default bench line
O=gpurun_out/${1:-r4chk}; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
VBX_SAMPLE_ADA_TABLE=0 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline > $O/sample_noada.json 2>/dev/null
python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline > $O/sample_ada.json 2>/dev/null
tail -6 $O/suite.log; tail -2 $O/smoke.log; python - <<PY
import json
for f in ("bench","sample_noada","sample_ada"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d.get("sample",{}).get("ms"))
    except Exception as e: print(f, "ERR", e)
PY
