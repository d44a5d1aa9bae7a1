cd $GRAFT_REPO_ROOT; O=gpurun_out/c16; mkdir -p $O
VBX_BM160_K64=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" > $O/pytest_k64.log 2>&1; tail -3 $O/pytest_k64.log
for k in 0 1 0 1; do
  echo "== K64 $k"; VBX_BM160_K64=$k timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "N=512" | cut -c1-100
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for k in 0 1 0 1; do
  VBX_BM160_K64=$k timeout 200 $B 2>/dev/null | tail -1 > $O/train_k$k.json
  python - $O/train_k$k.json $k <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("k64",sys.argv[2],"ms",d["ms_per_step"],"loss",d.get("final_loss"),{s:k.get(s) for s in ("fwd to_out","fwd ff_out","dgrad to_qkv","dgrad ff_in")})
PY
done
