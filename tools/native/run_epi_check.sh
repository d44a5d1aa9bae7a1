#!/bin/bash
# builds (if needed) and runs the stand-alone epilogue check against the in-tree libvbx_hip.so
D=$(cd "$(dirname "$0")" && pwd); R=$(cd "$D/../.." && pwd); L=$R/voicebox-pytorch_amd/lib
[ -x "$D/epi_check" ] || /opt/rocm/bin/hipcc -O1 -std=c++17 "$D/epi_check.cpp" -o "$D/epi_check" -L"$L" -lvbx_hip -Wl,-rpath,"$L" || exit 3
"$D/epi_check"
