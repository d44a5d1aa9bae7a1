#!/bin/bash
# last call of the round: full GPU suite + smoke on the final tree, kernel stats of the dim-1024 train step (BASELINE config 3)
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_d1024 -o run -- python $R/bench.py --dim 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-sample > $R/$O/prof_d1024.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_d1024 -name "*.db" | head -1) 9 > $O/r05_train_dim1024_kernel_stats.txt 2>&1
rm -rf $O/prof_d1024
head -14 $O/r05_train_dim1024_kernel_stats.txt
