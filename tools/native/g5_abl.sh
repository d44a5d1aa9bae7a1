#!/bin/bash
# timing ablations of gemm5 (VBX_G5_ABL bits: 1 no epilogue, 2 no DMA, 4 every workgroup reads row block 0, 8 no MFMAs)
cd "$(dirname "$0")/../.."
for a in ${@:-0 1 3 9 11}; do echo "ABL=$a"; VBX_G5_ABL=$a timeout 120 tools/native/gemm5_check time 2>&1 | grep -A4 "batch 8" | grep "to_qkv\|ff_in" | sed 's/128-wide.*gemm5/gemm5/'; done
