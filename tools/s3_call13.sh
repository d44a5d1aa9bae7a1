cd $GRAFT_REPO_ROOT; O=gpurun_out/c13; mkdir -p $O
for pf in 1 0; do
  VBX_BM160_PF=$pf timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_nt_bf16_f32 or gemm_nn" > $O/pytest_pf$pf.log 2>&1; echo "pf $pf:"; tail -2 $O/pytest_pf$pf.log
done
for pf in 0 1 0 1; do
  echo "== PF $pf"; VBX_BM160_PF=$pf timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "N=512" | cut -c1-100
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sample"
for pf in 0 1 0 1; do
  VBX_BM160_PF=$pf timeout 200 $B 2>/dev/null | tail -1 > $O/train_pf$pf.json
  python - $O/train_pf$pf.json $pf <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); k={x["stage"]:x["us_per_launch"] for x in d["roofline"]["kernels"]}
print("pf",sys.argv[2],"ms",d["ms_per_step"],{s:k.get(s) for s in ("fwd to_out","fwd ff_out","dgrad to_qkv","dgrad ff_in")})
PY
done
