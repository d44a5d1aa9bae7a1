#!/bin/bash
# usage: tools/bv.sh [bench args...]  -> prints "value ms_per_step" of one bench run (env vars select A/B variants)
python bench.py --no-cpu-baseline --no-sample "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3))"
