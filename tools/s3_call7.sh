cd $GRAFT_REPO_ROOT; O=gpurun_out/c7; mkdir -p $O
for s in 0 6 9 12 15; do VBX_GEMM_STAGGER=$s timeout 100 tools/native/gemm_trace time 2>&1 | tail -1; done > $O/stagger_time.txt
cat $O/stagger_time.txt
VBX_GEMM_STAGGER=12 timeout 100 tools/native/gemm_trace 3 > $O/trace_g4_stagger12.txt 2>&1
