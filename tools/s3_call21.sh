cd $GRAFT_REPO_ROOT; O=gpurun_out/c21; mkdir -p $O
VBX_V2_K64=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm or qkv or geglu" > $O/pytest_k64.log 2>&1; tail -2 $O/pytest_k64.log
for v in 0 1 0 1; do
  echo "== V2_K64 $v"; VBX_V2_K64=$v timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "to_qkv t|to_qkv e|ff_in t|ff_in e" | cut -c1-100
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for v in 0 1 0 1; do
  VBX_V2_K64=$v timeout 200 $B 2>/dev/null | tail -1 > $O/train_$v.json
  python - $O/train_$v.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("v2k64",sys.argv[2],"ms",d["ms_per_step"],"loss",d.get("final_loss"),{s:k.get(s) for s in ("fwd to_qkv","fwd ff_in")})
PY
done
for v in 0 1; do VBX_V2_K64=$v VBX_GEMM4_FFIN=$((1-v)) timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --intervals 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('sample v2k64=$v ms',d['ms_per_step'],[(x['stage'],x['us_per_launch']) for x in d['roofline']['kernels']])"; done
