cd $GRAFT_REPO_ROOT; O=gpurun_out/c6; mkdir -p $O
timeout 300 tools/native/gemm3_check correct > $O/correct.txt 2>&1; tail -4 $O/correct.txt
timeout 300 tools/native/gemm3_check race > $O/race.txt 2>&1; tail -25 $O/race.txt
timeout 100 tools/native/gemm_trace time 2>&1 | tail -1
timeout 100 tools/native/gemm_trace > $O/trace.txt 2>&1
grep -E "^==|epilogue|k-loop" $O/trace.txt
