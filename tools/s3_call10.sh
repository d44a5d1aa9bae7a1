cd $GRAFT_REPO_ROOT; O=gpurun_out/c10; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/sample_concurrent.py 16 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for v in 0 1 3 0 3; do
  VBX_GEMM_AUTO2=$v timeout 200 $B 2>/dev/null | tail -1 > $O/train_auto$v.json
  python - $O/train_auto$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("auto2",sys.argv[2],"ms",d["ms_per_step"],{s:k.get(s) for s in ("fwd ff_in","dgrad ff_out","dgrad to_out","fwd to_qkv")})
PY
done
