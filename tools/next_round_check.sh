# First GPU call of the next round: validate and A/B the experiments that were written after round 1's GPU time ran out.
# (one gpurun call, ~3 min; every step bounded)   usage: gpurun --timeout 400 -- 'bash tools/next_round_check.sh'
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# 0. the default path as committed (epilogue hoists were only checked by tools/native/epi_check on the GPU)
timeout 120 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
# 1. grouped weight-gradient GEMM launch
VBX_TEST_EXPERIMENTAL=1 timeout 60 python -m pytest tests/test_ops_gpu.py -m gpu -q -k grouped 2>&1 | tail -3
VBX_GROUP_WGRAD=1 timeout 120 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
  echo base $(tools/bv.sh)
  echo grouped $(VBX_GROUP_WGRAD=1 tools/bv.sh)
  echo grouped_s3 $(VBX_GROUP_WGRAD=1 VBX_WGRAD_SPLITS=3 tools/bv.sh)
  echo grouped_s2 $(VBX_GROUP_WGRAD=1 VBX_WGRAD_SPLITS=2 tools/bv.sh)
done
# 2. forward attention with the ragged-tile role (needs the experimental build)
VBX_BUILD_EXPERIMENTAL=1 python voicebox-pytorch_amd/build.py --force > gpurun_out/build_exp.log 2>&1
VBX_ATTN_RAGGED=1 timeout 60 python -m pytest tests/test_ops_gpu.py -m gpu -q -k attn_fwd 2>&1 | tail -3
for i in 1 2; do
  echo sample_base $(tools/bv.sh --mode sample --steps 3 --warmup 1)
  echo sample_ragged $(VBX_ATTN_RAGGED=1 tools/bv.sh --mode sample --steps 3 --warmup 1)
done
