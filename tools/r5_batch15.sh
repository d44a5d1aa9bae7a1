#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b15; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -q > $O/dp.log 2>&1; tail -3 $O/dp.log
