# session-3 call 1: validate HEAD, baseline bench, attention backward A/B (in-situ train step + stand-alone)
cd $GRAFT_REPO_ROOT; O=gpurun_out/c1; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for v in 4 0 3; do
  VBX_ATTN_BWD_DMA=$v timeout 200 $B 2>/dev/null | tail -1 > $O/train_dma$v.json
  python - $O/train_dma$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("dma",sys.argv[2],"ms",d["ms_per_step"],"attn_bwd",k.get("bwd attention"),"attn_fwd",k.get("fwd attention"))
PY
done
timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/sample.json; python -c "
import json;d=json.loads(open('$O/sample.json').read());print('sample ms',d['ms_per_step'],[(x['stage'],x['us_per_launch']) for x in d['roofline']['kernels']])"
for v in 0 3 4; do echo "attn_bench dma=$v"; VBX_ATTN_BWD_DMA=$v timeout 100 python tools/attn_bench.py 30 2>&1 | tail -2; done
