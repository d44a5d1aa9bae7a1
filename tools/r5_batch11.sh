#!/bin/bash
# precise-mode coverage + the kernels it touched
cd $GRAFT_REPO_ROOT; O=gpurun_out/b11; mkdir -p $O
timeout 900 python -m pytest tests/test_precise_gpu.py -q -x -s -k "text or gateloop or dropout or small_golden" > $O/precise.log 2>&1; tail -25 $O/precise.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -k "dropout or text or gateloop" > $O/touched.log 2>&1; tail -5 $O/touched.log
