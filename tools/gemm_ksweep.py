"""K sweep of the wide forward GEMM shapes: T(K) = a + b*K separates epilogue/launch cost (a) from the main loop (b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import gemm, L

M = 8320
for K in (64, 256, 512, 1024, 2048):
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, M, 3072, K, f16=1, name=f"NT f32-out N=3072 K={K}")
for K in (64, 256, 512, 1024, 2048):
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_BF16, M, 3072, K, f16=0, name=f"NT bf16-out N=3072 K={K}")
for K in (64, 256, 512, 1024, 2048):
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, M, 2816, K, f16=1, name=f"NT geglu N=2816 K={K}")
for K in (64, 512, 1024, 2048):
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, M, 512, K, f16=1, name=f"NT f32-out N=512 K={K}")
