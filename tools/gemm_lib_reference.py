"""Context for the roofline fractions: what the vendor GEMM library (torch.mm -> hipBLASLt / rocBLAS, no epilogue) reaches on this box for the plain
products of one layer at the benchmark shape (M = 8 x 1040 rows, dim 512).  Not part of the product path; back-to-back launches, L2-warm.
Usage: python tools/gemm_lib_reference.py [iters]"""
import sys, torch

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = 8 * 1040
shapes = [("fwd to_qkv   NT", M, 3072, 512, torch.float16), ("fwd ff_in    NT", M, 2816, 512, torch.float16),
          ("fwd to_out   NT", M, 512, 1024, torch.float16), ("fwd ff_out   NT", M, 512, 1408, torch.float16),
          ("dgrad to_qkv NN", M, 512, 3072, torch.bfloat16), ("dgrad ff_in  NN", M, 512, 2816, torch.bfloat16),
          ("dgrad to_out NN", M, 1024, 512, torch.bfloat16), ("dgrad ff_out NN", M, 1408, 512, torch.bfloat16),
          ("wgrad to_qkv TN", 3072, 512, M, torch.bfloat16), ("wgrad ff_in  TN", 2816, 512, M, torch.bfloat16),
          ("big square     ", 8192, 8192, 8192, torch.bfloat16)]
for name, m, n, k, dt in shapes:
    a = torch.randn(m, k, device=dev).to(dt)
    kind = name.split()[-1]
    if kind == "NT":
        w = torch.randn(n, k, device=dev).to(dt); f = lambda: torch.mm(a, w.t())
    elif kind == "TN":
        at = torch.randn(k, m, device=dev).to(dt); b = torch.randn(k, n, device=dev).to(dt); f = lambda: torch.mm(at.t(), b)
    else:
        b = torch.randn(k, n, device=dev).to(dt); f = lambda: torch.mm(a, b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name}  M={m:5d} N={n:5d} K={k:5d} {str(dt)[6:]:9s} {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
