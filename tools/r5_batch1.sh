#!/bin/bash
# Round-5 GPU batch 1: parity suite on the new attention contract + folded backward, A/B timings, step bench, probes.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
for f in 1 0; do
  echo "== VBX_ATTN_BWD_FOLD=$f" >> $O/attn_bench.txt
  VBX_ATTN_BWD_FOLD=$f timeout 300 python tools/attn_bench.py 50 >> $O/attn_bench.txt 2>&1
  VBX_ATTN_BWD_FOLD=$f timeout 300 python tools/attn_bench.py 50 >> $O/attn_bench.txt 2>&1
done
timeout 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err
VBX_ATTN_BWD_FOLD=0 timeout 600 python bench.py --no-cpu-baseline --no-sample > $O/bench_train_nofold.json 2> $O/bench_train_nofold.err
timeout 300 python tools/find_fills.py > $O/fills.txt 2>&1
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/mfma_long_power.txt 2>&1 &
SM=$!
timeout 300 tools/probes/mfma_long > $O/mfma_long.txt 2>&1
kill $SM
tail -3 $O/suite.log; cat $O/attn_bench.txt; tail -c 600 $O/bench_train.json; cat $O/mfma_long.txt
