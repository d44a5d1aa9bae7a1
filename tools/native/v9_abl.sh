#!/bin/bash
# Timing ablations of the one-round 160 x 128 x 64 tile (gemm_kernel_bm160k64: to_out, FeedForward-out, the N = dim dgrads) -- WRONG results
# by construction.  lib/v9abl<n>/libvbx_hip.so = gemm.hip compiled -DVBX_V9_ABL=<n>: 1 no DMA behind the prologue, 2 no fragment reads,
# 4 no barrier, 8 no MFMAs, 64 no epilogue.      here: tools/native/v9_abl.sh build [n ...]     on the GPU: tools/native/v9_abl.sh run [n ...]
cd "$(dirname "$0")/../.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
mode=$1; shift
NS=${@:-1 2 4 8 64 3 11 75}
if [ "$mode" = build ]; then
  for n in $NS; do mkdir -p $L/v9abl$n; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVBX_V9_ABL=$n -c $C/gemm.hip -o $L/v9abl$n/gemm.o & done
  wait
  for n in $NS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/v9abl$n/libvbx_hip.so $L/api.o $L/v9abl$n/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o || exit 1
  done
  echo built
else
  [ -x tools/native/gemm3_check ] || tools/native/run_gemm3_check.sh none > /dev/null 2>&1
  for n in 0 $NS; do
    echo "ABL=$n"
    if [ $n = 0 ]; then P=$L; else P=$L/v9abl$n; fi
    LD_LIBRARY_PATH=$P:$LD_LIBRARY_PATH timeout 200 tools/native/gemm3_check time 2>&1 | grep "N=512" | cut -c1-92
  done
fi
