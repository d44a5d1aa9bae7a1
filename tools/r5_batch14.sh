#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/b14; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
