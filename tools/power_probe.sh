#!/bin/bash
# Samples GPU clock / power (rocm-smi) while bench.py runs: is the train step running into the power cap (DVFS)?
# usage (GPU box): bash tools/power_probe.sh [bench args...]   -> gpurun_out/power_probe.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=gpurun_out/power_probe.log; : > $L
rocm-smi --showmaxpower --showclocks --showpower 2>&1 | grep -v "^=\|^$" >> $L
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction\)|mclk" | tr '\n' ' ' ; echo; sleep 0.2; done ) >> $L 2>&1 &
SM=$!
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sample "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" >> $L
kill $SM
grep -c sclk $L; grep sclk $L | awk 'NR%5==0' | head -40
tail -2 $L
