#!/bin/bash
# builds (if needed) and runs the GEMM workgroup-timeline tool against the diagnostic library (tools/build_trace_lib.sh first)
D=$(cd "$(dirname "$0")" && pwd); R=$(cd "$D/../.." && pwd); L=$R/voicebox-pytorch_amd/lib
if [ ! -x "$D/gemm_trace" ] || [ "$D/gemm_trace.cpp" -nt "$D/gemm_trace" ]; then
  /opt/rocm/bin/hipcc -O1 -std=c++17 "$D/gemm_trace.cpp" -o "$D/gemm_trace" -L"$L" -lvbx_hip_trace -Wl,-rpath,"$L" || exit 3
fi
"$D/gemm_trace" "$@"
