#!/bin/bash
# Timing ablations of the two-body attention backward's dk/dv body (WRONG results by construction -- diagnostic builds only):
# bit 0: no L / delta statistics reads (LDS), bit 1: no exponentials (VALU).  Builds libvbx_hip_abl<n>.so next to the product library;
# on the GPU: for n in 0 1 2 3; do VBX_LIB_PATH=voicebox-pytorch_amd/lib/libvbx_hip_abl$n.so VBX_ATTN_BWD_DMA=2 python tools/attn_bench.py 50; done
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
for n in 0 1 2 3; do /opt/rocm/bin/hipcc $F -DVBX_ATTN_ABL_DKDV=$n -c $C/attn.hip -o $L/attn_abl$n.o & done
wait
for n in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_abl$n.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn_abl$n.o $L/norm.o $L/gateloop.o $L/ops.o $L/runtime.o
done
echo built
