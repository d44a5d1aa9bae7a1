#!/bin/bash
# round-4 GPU call: v4 forward attention correctness + A/B timing, precise-mode tests, probe re-run
O=gpurun_out/r4b; mkdir -p $O
python -m pytest tests/test_ops_gpu.py -q -x -k "attn" 2>&1 | tail -15 > $O/attn_tests.log
for v in 0 1 2 3; do echo "== VBX_ATTN_V4=$v" >> $O/attn_bench.log; VBX_ATTN_V4=$v python tools/attn_bench.py 30 >> $O/attn_bench.log 2>&1; done
python -m pytest tests/test_precise_gpu.py -q -s 2>&1 | grep -v "^$" | tail -40 > $O/precise.log
./tools/probes/mfma_chain > $O/r04_probe_mfma_chain.txt 2>&1
python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -8 > $O/model_tests.log
for v in 0 1; do VBX_ATTN_V4=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_v4_$v.json 2> $O/bench_v4_$v.err; done
tail -3 $O/attn_tests.log; cat $O/attn_bench.log; tail -3 $O/model_tests.log
