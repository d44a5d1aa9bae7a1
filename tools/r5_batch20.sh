#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b20; mkdir -p $O
for i in 1; do PYTHONHASHSEED=$i timeout 600 python -m pytest tests/test_dp_gpu.py -q -k "deferred or factor_mode" > $O/t$i.log 2>&1; tail -1 $O/t$i.log; grep -n "^E " $O/t$i.log | head -5; done
