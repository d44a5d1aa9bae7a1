#!/bin/bash
# Timing ablations of the to_qkv epilogue functor (WRONG results by construction): diagnostic libraries
# voicebox-pytorch_amd/lib/libvbx_hip_qkvabl<n>.so with gemm.hip compiled -DVBX_EPIQKV_ABL=<n> (1 no stores, 2 no 1/|x| sequence,
# 4 no rotary loads; sums combine).   here: bash tools/kdim_gemm_ablation.sh build     on the GPU: bash tools/kdim_gemm_ablation.sh run
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
NS="1 2 4 7"
if [ "$1" = build ]; then
  for n in $NS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVBX_EPIQKV_ABL=$n -c $C/gemm.hip -o $L/gemm_qkvabl$n.o & done
  wait
  for n in $NS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_qkvabl$n.so $L/api.o $L/gemm_qkvabl$n.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o
  done
  echo built
else
  python tools/kdim_gemm_bench.py 2>&1 | grep to_qkv
  for n in $NS; do echo "EPIQKV_ABL=$n"; VBX_LIB_PATH=$L/libvbx_hip_qkvabl$n.so python tools/kdim_gemm_bench.py 2>&1 | grep to_qkv; done
fi
