cd $GRAFT_REPO_ROOT; O=gpurun_out/c4; mkdir -p $O
for s in 0 3 5 7 9 12; do VBX_GEMM_STAGGER=$s timeout 100 tools/native/gemm_trace time 2>&1 | tail -1; done > $O/stagger_time.txt
cat $O/stagger_time.txt
VBX_GEMM_STAGGER=7 timeout 100 tools/native/gemm_trace 3 > $O/trace_g4_stagger7.txt 2>&1
