#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -s > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
timeout 200 python tools/adamw_diag.py > $O/adamw_diag.txt 2>&1
grep -n "passed\|failed\|^FAILED" $O/suite.log | tail -30; tail -12 $O/adamw_diag.txt
