#!/bin/bash
# here: tools/native/g5_trace.sh build    on the GPU: tools/native/g5_trace.sh run [train 0|1] [geglu 0|1]
cd "$(dirname "$0")/../.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $L/g5trace
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -DVBX_G5_TRACE -c $C/gemm5.hip -o $L/g5trace/gemm5.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/g5trace/libvbx_hip.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/g5trace/gemm5.o $L/attn.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o || exit 1
  /opt/rocm/bin/hipcc -O1 -std=c++17 tools/native/g5_trace.cpp -o tools/native/g5_trace -L$L/g5trace -lvbx_hip -Wl,-rpath,'$ORIGIN/../../voicebox-pytorch_amd/lib/g5trace' || exit 1
  echo built
else
  shift; tools/native/g5_trace "$@"
fi
