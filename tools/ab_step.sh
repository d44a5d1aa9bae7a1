#!/bin/bash
# In-situ A/B of one tuning knob: runs bench.py (train step, or the sampler for sample_split) once per value, twice interleaved, and
# prints ms per step -- the way every "measured" entry of DESIGN.md section 8 was taken (same box, same call).
#   usage (GPU box): bash tools/ab_step.sh <knob>        e.g.  gpurun --timeout 900 -- 'bash tools/ab_step.sh splits3'
#   knobs: adaln_all   VBX_ADALN_BWD_ALL   1 0    d(time_emb) of all layers in one launch vs the per-layer adaLN backward
#          defer       VBX_DEFER_REDUCE    1 0    partial-record reductions after layer 0 vs per layer
#          layer_red   VBX_LAYER_REDUCE    0 1    slab reduce fused into the layer's batched reduce (loser)
#          splits3     VBX_WGRAD_SPLITS3   3 4 5 6 7 8 2   K splits of the grouped weight-gradient launch
#          wgrad_strm  VBX_WGRAD_STREAM    0 1    weight gradients on a side stream (loser)
#          sumsq_fold  VBX_SUMSQ_FOLD      1 0    clip norm from the slab-reduce partials
#          attn_fold   VBX_ATTN_BWD_FOLD   1 0    softmax statistics folded into the MFMA accumulator (attention backward)
#          factors     VBX_ADALN_FACTORS   1 0    adaLN weight gradients in factor form vs materialised
#          delta       VBX_DELTA_FUSED     0 1    delta = rowsum(dO o O) in the to_out dgrad epilogue (loser)
#          sample_split VBX_SAMPLE_SPLIT   2 1    sampler: two concurrent half batches vs one stream
#          gemm5       VBX_GEMM5           1 0    weight-stationary to_qkv / FeedForward-in (train step)
#          gemm5_sample VBX_GEMM5          1 0    the same in the 64-interval sampler
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
case "$1" in
  adaln_all) V=VBX_ADALN_BWD_ALL; S="1 0";; defer) V=VBX_DEFER_REDUCE; S="1 0";; layer_red) V=VBX_LAYER_REDUCE; S="0 1";;
  splits3) V=VBX_WGRAD_SPLITS3; S="3 4 5 6 7 8 2";; wgrad_strm) V=VBX_WGRAD_STREAM; S="0 1";; sumsq_fold) V=VBX_SUMSQ_FOLD; S="1 0";;
  attn_fold) V=VBX_ATTN_BWD_FOLD; S="1 0";; factors) V=VBX_ADALN_FACTORS; S="1 0";; delta) V=VBX_DELTA_FUSED; S="0 1";;
  sample_split) V=VBX_SAMPLE_SPLIT; S="2 1";;
  gemm5|gemm5_sample) V=VBX_GEMM5; S="1 0";;
  *) sed -n 2,17p $0; exit 1;;
esac
ARGS="--steps 30 --warmup 8 --no-cpu-baseline --no-sample"
[ "$1" = gemm5_sample ] && ARGS="--mode sample --steps 3 --warmup 1 --no-cpu-baseline"
[ "$1" = sample_split ] && ARGS="--mode sample --steps 3 --warmup 1 --no-cpu-baseline"
for i in 1 2; do for s in $S; do
  env $V=$s timeout 300 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = {x['stage']: x['us_per_launch'] for x in d.get('roofline', {}).get('kernels', [])}
print('$V=$s', d['ms_per_step'], 'ms;', 'bwd attention', k.get('bwd attention'), 'wgrad', k.get('wgrad (4 GEMMs)'), 'to_qkv', k.get('fwd to_qkv'), 'ff_in', k.get('fwd ff_in'))"
done; done
