#!/bin/bash
# deferred all-layer partial-record reduce + all-layer adaLN d(time_emb): full GPU suite, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/b13; mkdir -p $O
for i in 1 2; do
for v in "1 1" "1 0" "0 0"; do set -- $v
VBX_ADALN_BWD_ALL=$1 VBX_DEFER_REDUCE=$2 timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-sample 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ada_all=$1 defer=$2', d['ms_per_step'])"
done; done 2>&1 | tee $O/ab.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
