import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import gemm, L
M = 8320
for K in (64, 512, 2048):
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_BF16, M, 3072, K, f16=0, name=f"NT bf16-out N=3072 K={K}")
    gemm(L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, M, 2816, K, f16=1, name=f"NT geglu N=2816 K={K}")
