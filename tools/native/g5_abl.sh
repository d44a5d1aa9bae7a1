#!/bin/bash
# Timing ablations of gemm5 (WRONG results by construction): diagnostic libraries lib/g5abl<n>/libvbx_hip.so = gemm5.hip compiled
# -DVBX_G5_ABL=<n> (bits: 1 no epilogue, 2 no activation DMA, 8 no MFMAs).
#   here: tools/native/g5_abl.sh build [n ...]      on the GPU: tools/native/g5_abl.sh run [n ...]
cd "$(dirname "$0")/../.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
mode=$1; shift
NS=${@:-1 8 9 11}
if [ "$mode" = build ]; then
  for n in $NS; do mkdir -p $L/g5abl$n; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -DVBX_G5_ABL=$n -c $C/gemm5.hip -o $L/g5abl$n/gemm5.o & done
  wait
  for n in $NS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/g5abl$n/libvbx_hip.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/g5abl$n/gemm5.o $L/attn.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o || exit 1
  done
  echo built
else
  [ -x tools/native/gemm5_check ] || tools/native/run_gemm5_check.sh none > /dev/null
  for n in 0 $NS; do
    echo "ABL=$n"
    if [ $n = 0 ]; then P=$L; else P=$L/g5abl$n; fi
    LD_LIBRARY_PATH=$P:$LD_LIBRARY_PATH timeout 120 tools/native/gemm5_check time 2>&1 | grep -A4 "batch 8" | grep "to_qkv\|ff_in" | sed 's/128-wide.*gemm5/gemm5/'
  done
fi
