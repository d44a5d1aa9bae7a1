#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b22; mkdir -p $O
timeout 600 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -q -k "gradient_norm or train_step or trainer or trajectory or adamw" > $O/t.log 2>&1; tail -4 $O/t.log
