#!/bin/bash
# builds (if needed) and runs the stand-alone gemm3 check against the in-tree libvbx_hip.so:  run_gemm3_check.sh [correct|race|time|all]
D=$(cd "$(dirname "$0")" && pwd); R=$(cd "$D/../.." && pwd); L=$R/voicebox-pytorch_amd/lib
if [ ! -x "$D/gemm3_check" ] || [ "$D/gemm3_check.cpp" -nt "$D/gemm3_check" ]; then
  /opt/rocm/bin/hipcc -O1 -std=c++17 "$D/gemm3_check.cpp" -o "$D/gemm3_check" -L"$L" -lvbx_hip -Wl,-rpath,"$L" || exit 3
fi
"$D/gemm3_check" "$@"
