#!/bin/bash
# quick loop for the one-pass attention backward: op tests, stand-alone timings (both variants), step-level timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/c2; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "attn" > $O/attn_tests.log 2>&1; echo "attn tests rc=$?"; tail -3 $O/attn_tests.log
VBX_ATTN_BWD_ONEPASS=1 timeout 120 python tools/attn_bench.py 30 2>&1 | tail -2
timeout 120 python tools/attn_bench.py 30 2>&1 | tail -1
VBX_LIB_PATH=voicebox-pytorch_amd/lib/libvbx_hip_trace.so timeout 200 python tools/attn_bwd1_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
