"""GEMM / attention micro-benchmarks at the benchmark shapes (dim 512, B=8, 1024 frames): TF/s per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from voicebox_pytorch_amd import _lib

dev = "cuda"
L = _lib
lib = L.lib()
st = L.current_stream()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm(mode, epi, M, N, K, f16=0, splits=1, name=""):
    if mode == L.VBX_GEMM_NT:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif mode == L.VBX_GEMM_NN:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    A = A.half() if f16 else A.bfloat16()
    B = B.half() if f16 else B.bfloat16()
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb = mode, epi, M, N, K, A.shape[1], B.shape[1]
    d.A, d.B, d.f16, d.splits = A.data_ptr(), B.data_ptr(), f16, splits
    keep = [A, B]
    if epi == L.VBX_EPI_F32:
        C = torch.empty(M, N, device=dev); d.C, d.ldc = C.data_ptr(), N; keep.append(C)
    elif epi == L.VBX_EPI_BF16:
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); d.C, d.ldc = C.data_ptr(), N; keep.append(C)
    elif epi == L.VBX_EPI_SPLITK:
        C = torch.empty(splits, M, N, device=dev); d.C = C.data_ptr(); keep.append(C)
    elif epi == L.VBX_EPI_GEGLU:
        C = torch.empty(M, N // 2, device=dev, dtype=torch.float16); bias = torch.zeros(N, device=dev)
        d.C, d.ldc, d.bias = C.data_ptr(), N // 2, bias.data_ptr(); keep += [C, bias]
    sec = timeit(lambda: lib.vbx_gemm(d, st))
    tf = 2.0 * M * N * K / sec / 1e12
    print(f"{name:28s} M={M:5d} N={N:5d} K={K:5d}  {sec*1e6:8.1f} us  {tf:7.1f} TF/s  ({tf/25:.1f}% of 2.5 PF)")
    return keep


if __name__ == "__main__":
    M = 8320
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "nt"):
        gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, M, 3072, 512, f16=1, name="NT f32-out (qkv shape)")
        gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, M, 512, 1024, f16=1, name="NT f32-out (out-proj)")
        gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, M, 512, 1408, f16=1, name="NT f32-out (ff2)")
        gemm(L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, M, 2816, 512, f16=1, name="NT geglu (ff1)")
        gemm(L.VBX_GEMM_NT, L.VBX_EPI_F32, 8192, 8192, 8192, f16=1, name="NT f32-out 8192^3")
    if which in ("all", "nn"):
        gemm(L.VBX_GEMM_NN, L.VBX_EPI_BF16, M, 512, 3072, name="NN (dhn1)")
        gemm(L.VBX_GEMM_NN, L.VBX_EPI_BF16, M, 512, 2816, name="NN (dhn2)")
        gemm(L.VBX_GEMM_NN, L.VBX_EPI_BF16, M, 1408, 512, name="NN (dg)")
        gemm(L.VBX_GEMM_NN, L.VBX_EPI_BF16, M, 1024, 512, name="NN (dO)")
    if which in ("all", "tn"):
        gemm(L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, 3072, 512, M, splits=8, name="TN (dWqkv)")
        gemm(L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, 2816, 512, M, splits=9, name="TN (dW1)")
        gemm(L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, 512, 1408, M, splits=16, name="TN (dW2)")
    if which in ("all", "attn"):
        B, H, Np = 8, 16, 1040
        q = torch.randn(B, H, Np, 64, device=dev); q = (q / q.norm(dim=-1, keepdim=True) * 8 * L.lib().vbx_attn_q_prescale(10.0)).half()  # q16 contract: include/vbx.h
        k = torch.randn(B, H, Np, 64, device=dev); k = (k / k.norm(dim=-1, keepdim=True) * 8).half()
        v = torch.randn(B, H, Np, 64, device=dev).half()
        out = torch.empty(B, Np, H * 64, device=dev, dtype=torch.float16)
        outb = torch.empty(B, Np, H * 64, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(B, H, Np, device=dev)
        sec = timeit(lambda: L.call("vbx_attn_fwd", q, k, v, None, out, outb, lse, B, H, Np, 10.0, st))
        fl = 4.0 * B * H * Np * Np * 64
        print(f"attn fwd  {sec*1e6:8.1f} us  {fl/sec/1e12:7.1f} TF/s")
        do = (torch.randn(B, Np, H * 64, device=dev) * 1e-3).bfloat16()
        delta = torch.empty(B, H, Np, device=dev); dq = torch.empty(B, H, Np, 64, device=dev); dk = torch.empty_like(dq)
        dv = torch.empty(B, Np, H * 64, device=dev, dtype=torch.bfloat16)
        qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
        sec = timeit(lambda: L.call("vbx_attn_bwd", q, k, qb, kb, vb, None, out, 1, do, lse, delta, dq, dk, dv, H * 64, B, H, Np, 10.0, None, st))
        print(f"attn bwd  {sec*1e6:8.1f} us  {2.5*fl/sec/1e12:7.1f} TF/s (algorithmic 2.5x fwd)")


def gemm_padded(M, N, K, pad, name=""):
    """NT fp16 GEMM with row strides K+pad (channel-conflict experiment)."""
    A = torch.randn(M, K + pad, device=dev).half()
    B = torch.randn(N, K + pad, device=dev).half()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb = L.VBX_GEMM_NT, L.VBX_EPI_BF16, M, N, K, K + pad, K + pad
    d.A, d.B, d.f16, d.C, d.ldc = A.data_ptr(), B.data_ptr(), 1, C.data_ptr(), N
    sec = timeit(lambda: lib.vbx_gemm(d, st))
    tf = 2.0 * M * N * K / sec / 1e12
    print(f"{name:28s} M={M:5d} N={N:5d} K={K:5d} pad={pad:3d} {sec*1e6:8.1f} us  {tf:7.1f} TF/s")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "pad":
    for pad in (0, 8, 32, 64, 72, 0):
        gemm_padded(8320, 3072, 512, pad, "NT bf16-out qkv shape")
    for pad in (0, 64, 72):
        gemm_padded(8192, 8192, 8192, pad, "NT bf16-out 8192^3")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "norm":
    B, Np, D = 8, 1040, 512
    x = torch.randn(B, Np, D, device=dev)
    ada = torch.randn(B, 4 * D, device=dev)
    y16 = torch.empty(B, Np, D, device=dev, dtype=torch.float16)
    yb = torch.empty(B, Np, D, device=dev, dtype=torch.bfloat16)
    sec = timeit(lambda: L.call("vbx_rmsnorm_fwd", x, ada, ada[:, D:], 4 * D, None, y16, B, Np, 0, Np, D, st), iters=50)
    print(f"rmsnorm fwd (eval: fp16 out)    {sec*1e6:7.2f} us  {(x.numel()*4 + y16.numel()*2)/sec/1e12:5.2f} TB/s")
    sec = timeit(lambda: L.call("vbx_rmsnorm_fwd", x, ada, ada[:, D:], 4 * D, yb, y16, B, Np, 0, Np, D, st), iters=50)
    print(f"rmsnorm fwd (train: fp16+bf16)  {sec*1e6:7.2f} us  {(x.numel()*4 + y16.numel()*4)/sec/1e12:5.2f} TB/s")
