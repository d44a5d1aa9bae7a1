#!/bin/bash
# Same-call A/B of compile-time forward-attention variants (no stamps): lib/libvbx_hip_fv_<tag>.so = attn.hip with the given -D flags.
#   here: tools/attn_fwd_variants.sh build "<tag>:<flags>" ...     on the GPU: tools/attn_fwd_variants.sh run <tag> ...   (NP=<n>)
cd "$(dirname "$0")/.."
L=voicebox-pytorch_amd/lib; C=voicebox-pytorch_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  for v in "$@"; do tag=${v%%:*}; fl=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $fl -c $C/attn.hip -o $L/attn_fv_$tag.o &
  done; wait
  for v in "$@"; do tag=${v%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libvbx_hip_fv_$tag.so $L/api.o $L/gemm.o $L/gemm3.o $L/gemm4.o $L/gemm5.o $L/attn_fv_$tag.o $L/norm.o $L/gateloop.o $L/ops.o $L/precise.o $L/runtime.o && echo built $tag
  done
else
  for rep in 1 2 3; do for tag in "$@"; do
    echo "$tag NP=${NP:-1040}: $(VBX_LIB_PATH=$L/libvbx_hip_fv_$tag.so python tools/attn_bench.py 50 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done; done
fi
