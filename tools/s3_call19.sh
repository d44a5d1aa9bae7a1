cd $GRAFT_REPO_ROOT; O=gpurun_out/c19; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for v in "0 0" "128 1" "0 0" "128 1"; do set -- $v
  echo "== ABL $1 LATE $2"; VBX_GEMM_ABL=$1 VBX_GEMM_LATE_DMA=$2 timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "to_qkv|ff_in|dgrad ff_out|dgrad to_out" | cut -c1-150
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for v in "0 0" "128 0" "0 1" "0 0" "128 0" "0 1"; do set -- $v
  VBX_GEMM_ABL=$1 VBX_GEMM_LATE_DMA=$2 timeout 200 $B 2>/dev/null | tail -1 > $O/train_$1_$2.json
  python - $O/train_$1_$2.json "$1 $2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("abl/late",sys.argv[2],"ms",d["ms_per_step"],"loss",d.get("final_loss"),{s:k.get(s) for s in ("fwd to_qkv","fwd ff_in","dgrad ff_out","dgrad to_out")})
PY
done
