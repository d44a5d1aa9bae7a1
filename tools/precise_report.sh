#!/bin/bash
# precise-mode parity report (every printed line of the model-level precise tests) and its cost -> gpurun_out/<tag>/${T}_precise_parity.txt
T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_precise_gpu.py -m gpu -q -s -k "small_golden or cfg4 or cfg3 or text or gateloop or dropout" 2>&1 | sed -e 's/^[.F]*//' | grep -v "Warning\|warnings\|^$\|assert abs\|Docs:\|amdgpu.ids\|Consider using\|dl = abs" > $O/${T}_precise_parity.txt
python tools/precise_cost.py 2>&1 | grep -v amdgpu.ids >> $O/${T}_precise_parity.txt
python tools/precise_cost.py --dim 1024 --batch 2 2>&1 | grep -v amdgpu.ids >> $O/${T}_precise_parity.txt
cat $O/${T}_precise_parity.txt
