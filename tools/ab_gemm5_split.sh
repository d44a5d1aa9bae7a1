#!/bin/bash
# Sampler A/B (same box, same call): weight-stationary to_qkv / FeedForward-in on / off x two concurrent half batches / one stream,
# and the CU share each part's gemm5 launches get in split mode (VBX_GEMM5_CUS; default = CUs / 2).
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
run() { env "$@" timeout 300 python bench.py --mode sample --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  run VBX_GEMM5=0 VBX_SAMPLE_SPLIT=2
  run VBX_GEMM5=1 VBX_SAMPLE_SPLIT=2
  run VBX_GEMM5=1 VBX_SAMPLE_SPLIT=2 VBX_GEMM5_CUS=96
  run VBX_GEMM5=1 VBX_SAMPLE_SPLIT=2 VBX_GEMM5_CUS=160
  run VBX_GEMM5=1 VBX_SAMPLE_SPLIT=2 VBX_GEMM5_CUS=256
  run VBX_GEMM5=1 VBX_SAMPLE_SPLIT=1
  run VBX_GEMM5=0 VBX_SAMPLE_SPLIT=1
done
