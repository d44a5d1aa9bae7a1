cd $GRAFT_REPO_ROOT
export VBX_BM160_K64=1
VBX_BM160_ORDER=2 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_nt_bf16_f32 or gemm_nn" 2>&1 | tail -2
for o in 1 2 1 2; do
  echo "== ORDER $o"; VBX_BM160_ORDER=$o timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "N=512" | cut -c1-100
done
