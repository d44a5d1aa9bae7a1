cd $GRAFT_REPO_ROOT; O=gpurun_out/c20; mkdir -p $O
timeout 300 tools/native/gemm3_check correct > $O/correct.txt 2>&1; tail -1 $O/correct.txt
timeout 300 tools/native/gemm3_check race > $O/race.txt 2>&1; tail -1 $O/race.txt
for v in 0 1; do
  echo "== BM160ALL $v"; if [ $v = 1 ]; then export VBX_GEMM_BM160ALL=1; fi; timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "to_qkv|ff_in|dgrad ff_out|dgrad to_out" | cut -c1-100
done
