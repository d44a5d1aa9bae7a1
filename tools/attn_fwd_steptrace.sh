#!/bin/bash
# Per-tile time line of the forward attention (diagnostic library: attn.hip compiled -DVBX_ATTN_STEPTRACE -> lib/libvbx_hip_steptrace.so).
#   here: tools/attn_fwd_steptrace.sh build        on the GPU: tools/attn_fwd_steptrace.sh run   (NP=<n>, DMAPOS="0 1 2 3": where v3 issues the next tile's DMA)
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
if [ "$1" = build ]; then
  for d in ${DMAPOS:-0}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $XFLAGS -DVBX_ATTN_STEPTRACE -DVBX_FWD_DMAPOS=${d%%a*} -DVBX_FWD_ABL=$(echo $d | sed "s/^[0-9]*a\{0,1\}//;s/^$/0/") -c $C/attn.hip -o $L/attn_steptrace$d.o &
  done
  wait
  for d in ${DMAPOS:-0}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_steptrace$d.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn_steptrace$d.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o
    echo built $L/libvbx_hip_steptrace$d.so
  done
else
  for d in ${DMAPOS:-0}; do echo "== DMAPOS=$d"; VBX_LIB_PATH=$L/libvbx_hip_steptrace$d.so python tools/attn_fwd_steptrace.py 2>&1 | grep -v amdgpu.ids; done
fi
