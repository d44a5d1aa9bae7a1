#!/bin/bash
# Timing ablations of the forward attention step (WRONG results by construction -- diagnostic build only).  Builds
# voicebox-pytorch_amd/lib/libvbx_hip_diag.so = the product library with attn.hip compiled -DVBX_ATTN_DIAG (extra instantiations of the
# forward body, attn_fwd_v3_body.inc ABL bits: 1 no exponentials, 2 half of the P.V MFMAs, 4 a quarter of the S MFMAs, 8 no fragment
# reads, 16 no O rescale, 32 no row maximum, 64 no s_setprio, 128 no tile DMA / vmcnt wait, 256 no barrier).
#   here:        bash tools/attn_fwd_ablation.sh build
#   on the GPU:  bash tools/attn_fwd_ablation.sh run     (VBX_FWD_ABL3=<bits> selects an instantiation; 0 = the product kernel)
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVBX_ATTN_DIAG -c $C/attn.hip -o $L/attn_diag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_diag.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn_diag.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o
  echo built $L/libvbx_hip_diag.so
else
  for rep in 1 2; do for n in ${ABLS:-0 1 6 8 14 16 49 64 384 63 447}; do
    echo -n "ABL=$n  "; VBX_LIB_PATH=$L/libvbx_hip_diag.so VBX_FWD_ABL3=$n NP=${NP:-1040} python tools/attn_bench.py 50 2>&1 | grep "fwd_eval"
  done; done
fi
