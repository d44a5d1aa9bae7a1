#!/bin/bash
# builds (if needed) and runs the stand-alone gemm5 check against the in-tree libvbx_hip.so:  run_gemm5_check.sh [correct|race|time|all]
D=$(cd "$(dirname "$0")" && pwd); R=$(cd "$D/../.." && pwd); L=$R/voicebox-pytorch_amd/lib
if [ ! -x "$D/gemm5_check" ] || [ "$D/gemm5_check.cpp" -nt "$D/gemm5_check" ]; then
  /opt/rocm/bin/hipcc -O1 -std=c++17 "$D/gemm5_check.cpp" -o "$D/gemm5_check" -L"$L" -lvbx_hip -Wl,-rpath,"$L" || exit 3
fi
"$D/gemm5_check" "$@"
