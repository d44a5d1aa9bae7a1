#!/bin/bash
# Round 3, first GPU contact of the one-pass attention backward: its op tests first (bounded), then the whole GPU suite, stand-alone
# attention timings of both backward variants, and a short train bench with each.
cd $GRAFT_REPO_ROOT; O=gpurun_out/c1; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "attn" > $O/attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -15 $O/attn_tests.log
VBX_ATTN_BWD_ONEPASS=1 timeout 120 python tools/attn_bench.py 20 > $O/attn_bench_onepass.log 2>&1; tail -2 $O/attn_bench_onepass.log
timeout 120 python tools/attn_bench.py 20 > $O/attn_bench_twobody.log 2>&1; tail -2 $O/attn_bench_twobody.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "suite rc=$?"; tail -25 $O/pytest.log
grep -h "cfg4 reference-init\|cfg3 \|cfg5 B=8\|relative grad errors vs REFERENCE\|grad-norm rel errors vs REFERENCE" $O/pytest.log | cut -c1-1500
VBX_ATTN_BWD_ONEPASS=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample > $O/bench_onepass.log 2>&1; tail -1 $O/bench_onepass.log | cut -c1-600
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sample > $O/bench_twobody.log 2>&1; tail -1 $O/bench_twobody.log | cut -c1-600
