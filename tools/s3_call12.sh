cd $GRAFT_REPO_ROOT; O=gpurun_out/c12; mkdir -p $O
for pos in 1 3; do
  VBX_G3_DMAPOS=$pos timeout 300 tools/native/gemm3_check correct > $O/correct_pos$pos.txt 2>&1; echo "pos $pos correct:"; tail -2 $O/correct_pos$pos.txt
  VBX_G3_DMAPOS=$pos timeout 300 tools/native/gemm3_check race > $O/race_pos$pos.txt 2>&1; tail -1 $O/race_pos$pos.txt
done
for pos in 0 1 3; do
  echo "== DMAPOS $pos"; VBX_G3_DMAPOS=$pos timeout 300 tools/native/gemm3_check time 2>&1 | grep -E "layer wgrads, ONE|K sweep|\^3 NT bf16 path 2"
done
