#!/bin/bash
# Diagnostic library: attn.hip with -DVBX_ATTN_TRACE (per-workgroup start / loop-end / end timestamps), linked with the product's
# other objects -> voicebox-pytorch_amd/lib/libvbx_hip_trace.so.  Used by tools/attn_timeline.py through VBX_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DVBX_ATTN_TRACE -c $C/attn.hip -o $L/attn_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_trace.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/attn_trace.o $L/norm.o $L/gateloop.o $L/ops.o $L/runtime.o
echo built $L/libvbx_hip_trace.so
