#!/bin/bash
# Timing ablations of the folded attention backward's dk/dv body (WRONG results by construction).  Diagnostic libraries
# voicebox-pytorch_amd/lib/libvbx_hip_babl<n>.so = attn.hip compiled -DVBX_BWD_ABL=<n>: 1 exponentials out of the dP chain's gaps (2: and
# three quarters of the S / dP MFMAs + half of the dV / dK MFMAs gone), 8 no fragment / statistics reads, 32 no L / delta reads only,
# 128 no tile DMA / vmcnt wait, 256 no barrier.
#   here: bash tools/attn_bwd_ablation.sh build     on the GPU: bash tools/attn_bwd_ablation.sh run
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
NS="1 2 8 32 384 10"
if [ "$1" = build ]; then
  for n in $NS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVBX_BWD_ABL=$n -c $C/attn.hip -o $L/attn_babl$n.o & done
  wait
  for n in $NS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_babl$n.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn_babl$n.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o
  done
  echo built
else
  for rep in 1 2; do
    echo -n "ABL=0   "; python tools/attn_bench.py 30 2>&1 | grep bwd
    for n in $NS; do echo -n "ABL=$n   "; VBX_LIB_PATH=$L/libvbx_hip_babl$n.so python tools/attn_bench.py 30 2>&1 | grep bwd; done
  done
fi
