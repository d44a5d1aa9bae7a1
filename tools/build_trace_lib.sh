#!/bin/bash
# Diagnostic library: attn.hip with -DVBX_ATTN_TRACE and gemm3.hip / gemm4.hip with -DVBX_GEMM_TRACE (per-workgroup start /
# prologue-end / loop-end / end timestamps), linked with the product's other objects -> voicebox-pytorch_amd/lib/libvbx_hip_trace.so.
# Used by tools/attn_timeline.py (VBX_LIB_PATH) and tools/native/gemm_trace.cpp.
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value"
/opt/rocm/bin/hipcc $F -DVBX_ATTN_TRACE -c $C/attn.hip -o $L/attn_trace.o &
/opt/rocm/bin/hipcc $F -DVBX_GEMM_TRACE -c $C/gemm.hip -o $L/gemm_trace.o &
/opt/rocm/bin/hipcc $F -DVBX_GEMM_TRACE -c $C/gemm3.hip -o $L/gemm3_trace.o &
/opt/rocm/bin/hipcc $F -DVBX_GEMM_TRACE -c $C/gemm4.hip -o $L/gemm4_trace.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_trace.so $L/api.o $L/gemm_trace.o $L/gemm3_trace.o $L/gemm4_trace.o $L/attn_trace.o $L/norm.o $L/gateloop.o $L/ops.o $L/runtime.o
echo built $L/libvbx_hip_trace.so
