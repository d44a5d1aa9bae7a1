#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
VBX_ATTN_BWD_FOLD=0 timeout 300 python -m pytest tests/test_dp_gpu.py -m gpu -q -k adamw > $O/adamw_nofold.log 2>&1
tail -30 $O/suite.log; tail -5 $O/adamw_nofold.log
