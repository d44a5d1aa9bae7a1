#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b24; mkdir -p $O
timeout 60 python -m pytest tests/test_dp_gpu.py -q -k "gradient_norm or factor_mode or deferred" > $O/t.log 2>&1; tail -2 $O/t.log
