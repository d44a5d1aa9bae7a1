cd $GRAFT_REPO_ROOT; O=gpurun_out/c22; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 tools/native/gemm3_check correct > $O/correct.txt 2>&1; tail -1 $O/correct.txt
timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "to_qkv" | cut -c1-160
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for v in 1; do
  timeout 200 $B 2>/dev/null | tail -1 > $O/train_$v.json
  python - $O/train_$v.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("run",sys.argv[2],"ms",d["ms_per_step"],"loss",d.get("final_loss"),{s:k.get(s) for s in ("fwd to_qkv","fwd ff_in")})
PY
done
timeout 200 python bench.py --mode sample --steps 2 --warmup 1 --intervals 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('sample ms',d['ms_per_step'],[(x['stage'],x['us_per_launch']) for x in d['roofline']['kernels']])"
