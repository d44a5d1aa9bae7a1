#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/call4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
# ragged-tile role of the forward attention (round-1 experiment, never run): experimental library
export VBX_LIB_PATH=$GRAFT_REPO_ROOT/voicebox-pytorch_amd/lib/libvbx_hip_exp.so
VBX_ATTN_RAGGED=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attn_fwd" > $O/v3r_test.log 2>&1; tail -3 $O/v3r_test.log
for i in 1 2; do
  echo "sample exp-lib base   $(tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
  echo "sample exp-lib ragged $(VBX_ATTN_RAGGED=1 tools/bv.sh --mode sample --steps 3 --warmup 1)" | tee -a $O/summary.log
done
unset VBX_LIB_PATH
timeout 1500 bash tools/run_prof_r02.sh > $O/prof.log 2>&1; tail -5 $O/prof.log
