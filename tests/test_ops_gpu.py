"""GPU parity tests, op level: every C-ABI kernel entry point of libvbx_hip.so against a CPU fp32/fp64
PyTorch restatement of the reference op (oracle/restate.py) on the same seeded inputs.

Tolerances are stated per test.  bf16 operands carry 2^-9 relative rounding, so GEMM-like ops are
compared against a reference computed from the *same bf16-rounded inputs* in fp64 (isolates the
kernel's own error: fp32 accumulation order) with rtol 2e-3, and norms/elementwise at bf16 output
precision (rel 2^-8).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import restate

pytestmark = pytest.mark.gpu

dev = "cuda"


@pytest.fixture(scope="module")
def L():
    from voicebox_pytorch_amd import _lib

    _lib.lib()
    _lib.call("vbx_check_device", 0)
    return _lib


def st():
    return torch.cuda.current_stream().cuda_stream


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / ref.norm().clamp(min=1e-30))


def max_err(got, ref):
    return float((got.double().cpu() - ref.double().cpu()).abs().max())


# ----------------------------------------------------------------------------- hardware probes
def test_probe_tr16(L):
    """ds_read_b64_tr_b16 semantics assumed by gemm.hip/attn.hip: inside each 16-lane group, lane a
    receives element (a&3) of the 8 bytes addressed by lanes 4j + (a>>2), j=0..3."""
    data = torch.arange(4096, dtype=torch.int16, device=dev)
    g = torch.Generator().manual_seed(0)
    for trial in range(3):
        if trial == 0:
            off = torch.arange(64, dtype=torch.int32) * 4  # dense
        else:
            off = (torch.randperm(1024, generator=g)[:64] * 4).to(torch.int32)
        out = torch.zeros(256, dtype=torch.int16, device=dev)
        L.call("vbx_probe_tr16", data, off.to(dev), out, st())
        torch.cuda.synchronize()
        out = out.cpu().view(64, 4).to(torch.int64)
        exp = torch.zeros(64, 4, dtype=torch.int64)
        for l in range(64):
            grp, a = l // 16, l % 16
            for j in range(4):
                sup = grp * 16 + 4 * j + (a >> 2)
                exp[l, j] = int(off[sup]) + (a & 3)
        assert torch.equal(out, exp), f"trial {trial}\n got {out.tolist()}\n exp {exp.tolist()}"


@pytest.mark.parametrize("which,shape", [(0, (16, 32, 16)), (1, (32, 16, 32)), (2, (32, 16, 32))])
def test_probe_mfma_layout(L, which, shape):
    m, k, n = shape
    g = torch.Generator().manual_seed(which)
    a = torch.randint(-4, 5, (m, k), generator=g).float()
    b = torch.randint(-4, 5, (k, n), generator=g).float()  # asymmetric integers: exact in bf16/fp16
    c = torch.zeros(m, n, device=dev)
    L.call("vbx_probe_mfma", which, a.to(dev), b.to(dev), c, st())
    assert torch.equal(c.cpu(), a @ b)


# ----------------------------------------------------------------------------- GEMM
@pytest.fixture(params=[0, 2, 3], ids=["auto", "tile256x256", "tile128x256"])
def tile_path(request, L):
    """Every GEMM tile family behind vbx_gemm: the automatic choice, the 256 x 256 8-wave tile (gemm3.hip) and the 128 x 256 two-per-CU
    tile (gemm4.hip) wherever they can serve the descriptor -- both store through the row-staged epilogues of gemm_epi3.hpp."""
    L.lib().vbx_gemm_select(request.param)
    yield request.param
    L.lib().vbx_gemm_select(0)


def gemm(L, mode, epi, A, B, M, N, K, **kw):
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K = mode, epi, M, N, K
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.lda, d.ldb = A.shape[-1], B.shape[-1]
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    rc = L.lib().vbx_gemm(d, st())
    assert rc == 0, L.lib().vbx_last_error()


@pytest.mark.parametrize("M,N,K", [(300, 264, 200), (128, 128, 64), (8320, 512, 1024), (77, 1536, 512), (8320, 512, 1000), (8200, 512, 64), (4160, 512, 1408), (8320, 1024, 1024),
                                   (8320, 1024, 512), (1234, 1408, 512), (70, 576, 512)])
def test_gemm_nt_bf16_f32(L, M, N, K, tile_path):
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    Bw = bf(torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev)
    ref = A.double().cpu() @ Bw.double().cpu().t()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_BF16, A, Bw, M, N, K, C=out, ldc=N, bias=bias)
    assert rel_err(out, ref + bias.double().cpu()) < 4e-3  # bf16 output rounding
    out32 = torch.empty(M, N, device=dev)
    out_b = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_F32, A, Bw, M, N, K, C=out32, ldc=N, bias=bias, resid=resid, C2=out_b)
    full = ref + bias.double().cpu() + resid.double().cpu()
    assert rel_err(out32, full) < 1e-5
    assert rel_err(out_b, full) < 4e-3
    out32b = torch.empty(M, N, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_F32, A, Bw, M, N, K, C=out32b, ldc=N)
    assert rel_err(out32b, ref) < 1e-5
    # without a bias: at K = 512 and N a multiple of 64 the automatic choice is the weight-stationary kernel's plain bf16 epilogue
    out_nb = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_BF16, A, Bw, M, N, K, C=out_nb, ldc=N)
    assert rel_err(out_nb[:M], ref) < 4e-3 and bool(torch.isnan(out_nb[M]).all())


@pytest.mark.parametrize("M,N,K", [(300, 264, 200), (8320, 1024, 512), (130, 64, 1408), (8320, 512, 3072), (8320, 512, 360), (4160, 512, 1024), (8320, 1024, 3072)])
def test_gemm_nn(L, M, N, K, tile_path):
    """dgrad layout: C[M,N] = A[M,K] . B[K,N]  (B read through the hardware transpose path)."""
    g = torch.Generator().manual_seed(M * 3 + N)
    A = bf(torch.randn(M, K, generator=g)).to(dev)
    Bm = bf(torch.randn(K, N, generator=g) * K ** -0.5).to(dev)
    ref = A.double().cpu() @ Bm.double().cpu()
    out32 = torch.empty(M, N, device=dev)
    gemm(L, L.VBX_GEMM_NN, L.VBX_EPI_F32, A, Bm, M, N, K, C=out32, ldc=N)
    assert rel_err(out32, ref) < 1e-5
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NN, L.VBX_EPI_BF16, A, Bm, M, N, K, C=out, ldc=N)
    assert rel_err(out, ref) < 4e-3


@pytest.mark.parametrize("M,N,K,splits", [(264, 200, 300, 1), (256, 128, 8320, 8), (2816, 512, 1000, 3)])
def test_gemm_tn_splitk(L, M, N, K, splits):
    """wgrad layout: C[M,N] = A[K,M]^T . B[K,N], fp32 split-K slabs + deterministic reduce."""
    g = torch.Generator().manual_seed(M + K)
    A = bf(torch.randn(K, M, generator=g)).to(dev)
    Bm = bf(torch.randn(K, N, generator=g)).to(dev)
    ref = A.double().cpu().t() @ Bm.double().cpu()
    slabs = torch.empty(splits, M, N, device=dev)
    gemm(L, L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, A, Bm, M, N, K, C=slabs, splits=splits)
    dst = torch.full((M, N), 7.0, device=dev)
    L.call("vbx_splitk_reduce", slabs, splits, M, N, dst, M, N, N, 0, 0, 0, st())
    assert rel_err(dst, ref) < 1e-5
    # accumulate + column trim
    dst2 = torch.ones(M, N - 3, device=dev)
    L.call("vbx_splitk_reduce", slabs, splits, M, N, dst2, M, N - 3, N - 3, 0, 0, 1, st())
    assert rel_err(dst2, ref[:, : N - 3] + 1) < 1e-5


def test_gemm_tn_geglu_rowmap(L):
    Fd, D, K = 170, 64, 333  # packed rows 2*Fp with Fp = 192
    Fp = 192
    g = torch.Generator().manual_seed(1)
    A = bf(torch.randn(K, 2 * Fp, generator=g)).to(dev)
    Bm = bf(torch.randn(K, D, generator=g)).to(dev)
    ref_packed = A.double().cpu().t() @ Bm.double().cpu()
    slabs = torch.empty(2, 2 * Fp, D, device=dev)
    gemm(L, L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, A, Bm, 2 * Fp, D, K, C=slabs, splits=2)
    dst = torch.zeros(2 * Fd, D, device=dev)
    L.call("vbx_splitk_reduce", slabs, 2, 2 * Fp, D, dst, 2 * Fd, D, D, 1, Fd, 0, st())
    exp = torch.zeros(2 * Fd, D, dtype=torch.float64)
    for p in range(2 * Fp):
        blk, w = p // 128, p % 128
        f = blk * 64 + (w & 63)
        if f < Fd:
            exp[f if w < 64 else Fd + f] = ref_packed[p]
    assert rel_err(dst, exp) < 1e-5


def rot_tables(Np, R):
    pos = torch.cat((torch.full((R,), -10000, dtype=torch.long), torch.arange(Np - R)))
    fr = restate.rotary_freqs(pos, 64, 50000.0)
    return fr, fr[:, :32].cos().contiguous(), fr[:, :32].sin().contiguous()


@pytest.mark.parametrize("Bsz,Np,H,D,qknorm", [(2, 56, 2, 64, True), (2, 1040, 4, 256, True), (1, 40, 2, 128, False),
                                               # K = 512: the weight-stationary kernel (csrc/gemm5.hip) under "auto" -- ragged last block,
                                               # batches changing inside a block, a panel with two idle waves (H = 2), Np < 32, no qk-norm
                                               (3, 100, 4, 512, True), (2, 77, 2, 512, True), (1, 24, 6, 512, True), (2, 33, 2, 512, False),
                                               (4, 1040, 16, 512, True),
                                               # >= 64 row blocks: the XCD-aware work map; 7 x 1049 = 7343 rows: ragged last block, odd Np
                                               (7, 1049, 16, 512, True)])
def test_gemm_qkv_epilogue(L, Bsz, Np, H, D, qknorm, tile_path):
    """to_qkv + MultiheadRMSNorm + rotary fused (voicebox_pytorch.py:320-328)."""
    g = torch.Generator().manual_seed(Np)
    I = H * 64
    M = Bsz * Np
    x = torch.randn(M, D, generator=g).half().to(dev)  # the runtime feeds this projection fp16 operands
    W = (torch.randn(3 * I, D, generator=g) * D ** -0.5).half().to(dev)
    qg = (1 + 0.1 * torch.randn(H, 64, generator=g)).to(dev)
    kg = (1 + 0.1 * torch.randn(H, 64, generator=g)).to(dev)
    fr, rc, rs = rot_tables(Np, 16)
    q16 = torch.full((Bsz, H, Np, 64), float("nan"), dtype=torch.float16, device=dev)
    k16 = torch.full_like(q16, float("nan"))
    qb = torch.full((Bsz, H, Np, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    kb = torch.full_like(qb, float("nan"))
    v = torch.full_like(qb, float("nan"))
    v16 = torch.full_like(q16, float("nan"))
    qrn = torch.full((Bsz, H, Np), float("nan"), device=dev)
    krn = torch.full_like(qrn, float("nan"))
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_QKV, x, W, M, 3 * I, D, Np=Np, H=H, qk_scale=8.0 if qknorm else 0.0,
         q_gamma=qg, k_gamma=kg, rot_cos=rc.to(dev), rot_sin=rs.to(dev), q16=q16, k16=k16, qb=qb, kb=kb, v=v,
         q_rnorm=qrn, k_rnorm=krn, f16=1, v16=v16, q_prescale=L.lib().vbx_attn_q_prescale(10.0))
    qkv = (x.double().cpu() @ W.double().cpu().t()).view(Bsz, Np, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, vv = qkv[0], qkv[1], qkv[2]
    assert rel_err(qrn, 1 / q.norm(dim=-1)) < 1e-5 and rel_err(krn, 1 / k.norm(dim=-1)) < 1e-5
    if qknorm:
        q = restate.l2norm_scale(q, 64) * qg.double().cpu()[:, None, :]
        k = restate.l2norm_scale(k, 64) * kg.double().cpu()[:, None, :]
    q, k = restate.apply_rotary(fr.double(), q), restate.apply_rotary(fr.double(), k)
    # fp16 storage: 2^-11; q16 carries the attention kernels' scale * log2(e) (include/vbx.h), qb / k16 / kb do not
    assert rel_err(q16, q * L.lib().vbx_attn_q_prescale(10.0)) < 6e-4 and rel_err(k16, k) < 6e-4
    assert rel_err(qb, q) < 4e-3 and rel_err(kb, k) < 4e-3
    assert rel_err(v, vv) < 4e-3 and rel_err(v16, vv) < 6e-4


@pytest.mark.parametrize("Bsz,Np,H", [(3, 100, 4), (2, 1040, 16)])
def test_gemm5_inference_outputs_and_tiled_agreement(L, Bsz, Np, H):
    """K = 512 to_qkv without the backward's copies (what the sampler runs): the weight-stationary kernel writes q16 / k16 / v16 only,
    agrees with fp64, and agrees with the 128-wide tiled kernels to rounding (the two differ in fp32 summation order and in the
    1 / |x| sequence: v_rsq + Newton step against IEEE sqrt + divide)."""
    g = torch.Generator().manual_seed(Np + H)
    D, I, M = 512, H * 64, Bsz * Np
    x = torch.randn(M, D, generator=g).half().to(dev)
    W = (torch.randn(3 * I, D, generator=g) * D ** -0.5).half().to(dev)
    qg = (1 + 0.1 * torch.randn(H, 64, generator=g)).to(dev)
    kg = (1 + 0.1 * torch.randn(H, 64, generator=g)).to(dev)
    fr, rc, rs = rot_tables(Np, 16)
    ps = L.lib().vbx_attn_q_prescale(10.0)
    outs = {}
    for path in (0, 1):  # automatic (gemm5) / the 128-wide tiled kernels
        L.lib().vbx_gemm_select(path)
        q16 = torch.full((Bsz, H, Np, 64), float("nan"), dtype=torch.float16, device=dev)
        k16, v16 = torch.full_like(q16, float("nan")), torch.full_like(q16, float("nan"))
        gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_QKV, x, W, M, 3 * I, D, Np=Np, H=H, qk_scale=8.0, q_gamma=qg, k_gamma=kg, rot_cos=rc.to(dev),
             rot_sin=rs.to(dev), q16=q16, k16=k16, f16=1, v16=v16, q_prescale=ps)
        outs[path] = (q16, k16, v16)
    L.lib().vbx_gemm_select(0)
    qkv = (x.double().cpu() @ W.double().cpu().t()).view(Bsz, Np, 3, H, 64).permute(2, 0, 3, 1, 4)
    q = restate.apply_rotary(fr.double(), restate.l2norm_scale(qkv[0], 64) * qg.double().cpu()[:, None, :])
    k = restate.apply_rotary(fr.double(), restate.l2norm_scale(qkv[1], 64) * kg.double().cpu()[:, None, :])
    for path in (0, 1):
        q16, k16, v16 = outs[path]
        assert rel_err(q16, q * ps) < 6e-4 and rel_err(k16, k) < 6e-4 and rel_err(v16, qkv[2]) < 6e-4
    for a, b in zip(outs[0], outs[1]):  # at most an fp16 ulp apart, almost everywhere equal
        d = (a.float() - b.float()).abs()
        assert float((d / (b.float().abs() + 1e-3)).max()) < 2.5e-3 and float((d > 0).float().mean()) < 0.02


@pytest.mark.parametrize("M,Fd,Fp,train", [(300, 341, 384, True), (70, 60, 64, False), (2100, 1365, 1408, True), (1234, 1365, 1408, False),
                                           (7343, 1365, 1408, True)])
def test_gemm5_geglu(L, M, Fd, Fp, train):
    """FeedForward[0] + GEGLU at K = 512 on the weight-stationary kernel (fp16 operands, the runtime's forward): fp16 G (+ bf16 copy
    and the interleaved bf16 pre-activation in training), ragged M, against fp64."""
    g = torch.Generator().manual_seed(M)
    D = 512
    x = torch.randn(M, D, generator=g).half().to(dev)
    W1 = torch.randn(2 * Fd, D, generator=g) * D ** -0.5
    b1 = torch.randn(2 * Fd, generator=g) * 0.1
    W1p = torch.empty(2 * Fp, D, dtype=torch.bfloat16, device=dev)
    b1p = torch.empty(2 * Fp, device=dev)
    W1h = torch.empty(2 * Fp, D, dtype=torch.float16, device=dev)
    L.call("vbx_pack_weight", W1.to(dev), 2 * Fd, D, W1p, W1h, 2 * Fp, D, 1, Fd, st())
    L.call("vbx_pack_bias", b1.to(dev), 2 * Fd, b1p, 2 * Fp, 1, Fd, st())
    g16 = torch.full((M + 1, Fp), float("nan"), dtype=torch.float16, device=dev)
    gb = torch.full((M + 1, Fp), float("nan"), dtype=torch.bfloat16, device=dev)
    h1 = torch.full((M + 1, 2 * Fp), float("nan"), dtype=torch.bfloat16, device=dev)
    kw = dict(C2=h1, C3=gb) if train else {}
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, x, W1h, M, 2 * Fp, D, C=g16, ldc=Fp, bias=b1p, f16=1, **kw)
    hdn = x.double().cpu() @ W1.half().double().t() + b1.double()
    a, gate = hdn.chunk(2, dim=-1)
    ref = F.gelu(gate) * a
    assert rel_err(g16[:M, :Fd], ref) < 6e-4
    assert float(g16[:M, Fd:].float().abs().max()) == 0.0 and bool(torch.isnan(g16[M]).all())  # padding exactly zero, nothing past row M
    if train:
        assert rel_err(gb[:M, :Fd], ref) < 4e-3 and bool(torch.isnan(gb[M]).all()) and bool(torch.isnan(h1[M]).all())
        blk = h1[:M].float().cpu().view(M, Fp // 64, 2, 64)
        assert rel_err(blk[:, :, 0].reshape(M, Fp)[:, :Fd], a) < 4e-3
        assert rel_err(blk[:, :, 1].reshape(M, Fp)[:, :Fd], gate) < 4e-3
    else:
        assert bool(torch.isnan(gb).all()) and bool(torch.isnan(h1).all())  # inference: the backward's copies are not written


@pytest.mark.parametrize("Fd,Fp", [(341, 384), (405, 448)])
def test_gemm_geglu_epilogue(L, tile_path, Fd, Fp):
    """FeedForward[0] + GEGLU fused, packed/interleaved weights (voicebox_pytorch.py:338-345).  Fp = 448: 2 * Fp = 896 leaves a
    128-column tail tile on the 128 x 256 kernel (the dim-1024 model's 2 * Fp = 5504 does too): its `col0 >= N` branch."""
    g = torch.Generator().manual_seed(3)
    M, D = 200, 128
    x = bf(torch.randn(M, D, generator=g)).to(dev)
    W1 = torch.randn(2 * Fd, D, generator=g) * D ** -0.5
    b1 = torch.randn(2 * Fd, generator=g) * 0.1
    W1p = torch.empty(2 * Fp, D, dtype=torch.bfloat16, device=dev)
    b1p = torch.empty(2 * Fp, device=dev)
    W1h = torch.empty(2 * Fp, D, dtype=torch.float16, device=dev)
    L.call("vbx_pack_weight", W1.to(dev), 2 * Fd, D, W1p, W1h, 2 * Fp, D, 1, Fd, st())
    L.call("vbx_pack_bias", b1.to(dev), 2 * Fd, b1p, 2 * Fp, 1, Fd, st())
    gout = torch.empty(M, Fp, dtype=torch.bfloat16, device=dev)
    h1 = torch.empty(M, 2 * Fp, dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, x, W1p, M, 2 * Fp, D, C=gout, ldc=Fp, bias=b1p, C2=h1)
    hdn = x.double().cpu() @ bf(W1).double().t() + b1.double()
    a, gate = hdn.chunk(2, dim=-1)
    ref = F.gelu(gate) * a
    assert rel_err(gout[:, :Fd], ref) < 4e-3
    # fp16 operand mode (what the runtime's forward uses): fp16 G + bf16 copy
    g16 = torch.empty(M, Fp, dtype=torch.float16, device=dev)
    gb = torch.empty(M, Fp, dtype=torch.bfloat16, device=dev)
    gemm(L, L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, x.half(), W1h, M, 2 * Fp, D, C=g16, ldc=Fp, bias=b1p, C2=h1, C3=gb, f16=1)
    hdn16 = x.half().double().cpu() @ W1.half().double().t() + b1.double()
    a16, gate16 = hdn16.chunk(2, dim=-1)
    assert rel_err(g16[:, :Fd], F.gelu(gate16) * a16) < 6e-4
    assert rel_err(gb[:, :Fd], F.gelu(gate16) * a16) < 4e-3
    assert float(gout[:, Fd:].float().abs().max()) == 0.0  # padding columns are exactly zero
    # saved pre-activation is the interleaved layout
    blk = h1.float().cpu().view(M, Fp // 64, 2, 64)
    assert rel_err(blk[:, :, 0].reshape(M, Fp)[:, :Fd], a) < 4e-3
    assert rel_err(blk[:, :, 1].reshape(M, Fp)[:, :Fd], gate) < 4e-3


# ----------------------------------------------------------------------------- norms
@pytest.mark.parametrize("Bsz,Np,n0,rpb,D,adaptive", [(2, 40, 0, 40, 64, True), (8, 1040, 0, 1040, 512, True),
                                                     (2, 56, 16, 40, 256, False), (3, 33, 0, 33, 1024, True),
                                                     (2, 19, 0, 19, 2048, True)])
def test_rmsnorm_fwd_bwd(L, Bsz, Np, n0, rpb, D, adaptive):
    g = torch.Generator().manual_seed(D + Np)
    x = torch.randn(Bsz, Np, D, generator=g)
    if adaptive:
        gamma, beta = 1 + 0.3 * torch.randn(Bsz, D, generator=g), 0.3 * torch.randn(Bsz, D, generator=g)
        stride = D
    else:
        gamma, beta, stride = 1 + 0.3 * torch.randn(D, generator=g), None, 0
    y = torch.empty(Bsz * rpb, D, dtype=torch.bfloat16, device=dev)
    xd, gd = x.to(dev), gamma.to(dev)
    bd = beta.to(dev) if beta is not None else None
    y16 = torch.empty(Bsz * rpb, D, dtype=torch.float16, device=dev)
    L.call("vbx_rmsnorm_fwd", xd, gd, bd, stride, y, y16, Bsz, Np, n0, rpb, D, st())
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    br = beta.double().requires_grad_(True) if beta is not None else None
    xs = xr[:, n0:n0 + rpb]
    nrm = restate.l2norm_scale(xs, D)
    ref = nrm * (gr[:, None, :] if adaptive else gr) + (br[:, None, :] if adaptive else 0.0)
    assert rel_err(y.view(Bsz, rpb, D), ref) < 4e-3
    assert rel_err(y16.view(Bsz, rpb, D), ref) < 6e-4
    # backward
    dy = bf(torch.randn(Bsz, rpb, D, generator=g))
    ref.backward(dy.double())
    dx_in = torch.randn(Bsz, Np, D, generator=g)
    dx_out = torch.zeros(Bsz, Np, D, device=dev)
    dxb = torch.zeros(Bsz, Np, D, dtype=torch.bfloat16, device=dev)
    chunks = L.lib().vbx_rmsnorm_bwd_chunks(rpb)
    part = torch.zeros(Bsz, chunks, 2, D, device=dev)
    cpart = torch.zeros(Bsz, chunks, D, device=dev)
    L.call("vbx_rmsnorm_bwd", xd, gd, stride, dy.to(dev), dx_in.to(dev), dx_out, dxb, part, cpart, Bsz, Np, n0, rpb, D, st())
    csum = torch.zeros(D, device=dev)
    L.call("vbx_reduce_col_partials", cpart, csum, torch.zeros(Bsz, D, device=dev), Bsz, chunks, D, st())
    assert rel_err(csum, dx_in[:, n0:n0 + rpb].double().sum((0, 1))) < 1e-5  # fused column sums of dx_in
    exp_dx = xr.grad[:, n0:n0 + rpb] + dx_in[:, n0:n0 + rpb].double()
    assert rel_err(dx_out[:, n0:n0 + rpb], exp_dx) < 1e-5
    assert rel_err(dxb[:, n0:n0 + rpb], exp_dx) < 4e-3
    if adaptive:
        out = torch.zeros(Bsz, 2, D, device=dev)
        L.call("vbx_reduce_norm_partials", part, out, 2 * D, Bsz, chunks, D, 0, st())
        assert rel_err(out[:, 0], gr.grad) < 1e-5 and rel_err(out[:, 1], br.grad) < 1e-5
    else:
        out = torch.zeros(2, D, device=dev)
        L.call("vbx_reduce_norm_partials", part, out, 0, Bsz, chunks, D, 1, st())
        assert rel_err(out[0], gr.grad) < 1e-5


# ----------------------------------------------------------------------------- attention
def attn_scratch(L, Bsz, H, Np):
    """The `scratch` argument of vbx_attn_bwd*: no kernel uses it since round 6 (vbx_attn_bwd_scratch_bytes() == 0); kept in the ABI."""
    assert L.lib().vbx_attn_bwd_scratch_bytes(Bsz, H, Np) == 0
    return None


@pytest.fixture(params=[3, 1], ids=["unfolded", "folded"])
def bwd_variant(request, L):
    """Both attention-backward kernels behind vbx_attn_bwd / vbx_attn_bwd_fused: the two-body kernel with the softmax statistics folded
    into the MFMA accumulator (round 5, default) and the same bodies without the fold (select 3; also what attention dropout runs on).
    The one-pass chain kernel of round 3 (select 2) was removed in round 6: selecting it is an error."""
    assert L.lib().vbx_attn_bwd_select(2) != 0
    L.lib().vbx_attn_bwd_select(request.param)
    yield request.param
    L.lib().vbx_attn_bwd_select(0)


def qpre(L, q16, scale):
    """The kernels' q operand and the q the exact reference must see (include/vbx.h, attention contract since round 5): q16 carries
    scale * log2(e), i.e. the kernel computes with fp16(q * c) -- the reference with that value divided by c in fp64."""
    c = L.lib().vbx_attn_q_prescale(scale)
    qs = (q16.float() * c).half()
    return qs, qs.double() / c


def attn_inputs(Bsz, H, Np, seed, qnorm=8.0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(Bsz, H, Np, 64, generator=g)
    k = torch.randn(Bsz, H, Np, 64, generator=g)
    if qnorm:
        q = q / q.norm(dim=-1, keepdim=True) * qnorm
        k = k / k.norm(dim=-1, keepdim=True) * qnorm
    v = torch.randn(Bsz, H, Np, 64, generator=g)
    return q.half(), k.half(), v.half()


@pytest.mark.parametrize("Bsz,H,Np,scale,masked", [(1, 2, 64, 10.0, False), (2, 2, 1040, 10.0, False),
                                                   (2, 3, 77, 10.0, True), (1, 2, 200, 0.125, True),
                                                   (2, 2, 1040, 10.0, True), (1, 2, 128, 10.0, False), (2, 2, 300, 10.0, True),
                                                   (1, 3, 24, 10.0, False),
                                                   # ragged tail tiles of <= 16 query rows (the forward's 16 x 16 MFMA role, round 6)
                                                   (2, 2, 144, 10.0, True), (1, 3, 130, 10.0, False), (2, 2, 264, 10.0, True),
                                                   (1, 2, 16, 10.0, False), (1, 2, 1029, 10.0, True)])
def test_attn_fwd_bwd(L, Bsz, H, Np, scale, masked, bwd_variant):
    """Attend.forward math path (attend.py:121-135) at the reference's logit scale (10 * q.k, |q|=|k|=8)."""
    q16, k16, v = attn_inputs(Bsz, H, Np, seed=Np + H, qnorm=8.0 if scale == 10.0 else None)
    mask = None
    if masked:
        mask = torch.ones(Bsz, Np, dtype=torch.bool)
        mask[0, Np - 13:] = False
        if Bsz > 1:
            mask[1, 5:9] = False
    out16 = torch.empty(Bsz, Np, H * 64, dtype=torch.float16, device=dev)
    out = torch.empty(Bsz, Np, H * 64, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(Bsz, H, Np, device=dev)
    qs, q_eff = qpre(L, q16, scale)
    qd, kd, vd = qs.to(dev), k16.to(dev), v.to(dev)
    md = mask.to(dev) if masked else None
    L.call("vbx_attn_fwd", qd, kd, vd, md, out16, out, lse, Bsz, H, Np, scale, st())
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q_eff, k16, v))
    ref = restate.attend(qr, kr, vr, mask=mask, scale=scale)  # (b,h,n,d)
    ref_t = ref.permute(0, 2, 1, 3).reshape(Bsz, Np, H * 64)
    # P is rounded to fp16 before P.V (rel 2^-11 per weight); outputs stored in fp16 (+ bf16 copy)
    assert rel_err(out16, ref_t) < 1.5e-3, rel_err(out16, ref_t)
    assert rel_err(out, ref_t) < 5e-3, rel_err(out, ref_t)
    sim = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    if masked:
        sim = sim.masked_fill(~mask[:, None, None, :], -float("inf"))
    ref_lse = torch.logsumexp(sim, dim=-1) / math.log(2.0)
    assert max_err(lse, ref_lse) < 2e-3
    # ---- backward
    g = torch.Generator().manual_seed(5)
    dout = bf(torch.randn(Bsz, Np, H * 64, generator=g) * 1e-3)  # gradient-sized values (fp16 would flush these)
    ref_t.backward(dout.double())
    qb, kb = bf(q16.float()).to(dev), bf(k16.float()).to(dev)
    delta = torch.empty(Bsz, H, Np, device=dev)
    dq = torch.zeros(Bsz, H, Np, 64, device=dev)
    dk = torch.zeros(Bsz, H, Np, 64, device=dev)
    dv = torch.zeros(Bsz, Np, 3 * H * 64, dtype=torch.bfloat16, device=dev)
    dv_view = dv[:, :, 2 * H * 64:]
    L.call("vbx_attn_bwd", qd, kd, qb, kb, bf(v.float()).to(dev), md, out16, 1, dout.to(dev), lse, delta, dq, dk,
           dv_view.data_ptr(), 3 * H * 64, Bsz, H, Np, scale, attn_scratch(L, Bsz, H, Np), st())
    torch.cuda.synchronize()
    dv_got = dv_view.float().cpu().view(Bsz, Np, H, 64).permute(0, 2, 1, 3)
    # operands of the backward GEMMs (P, dS, q, k, dO) are bf16 -> ~1e-2 relative on the result
    assert rel_err(dv_got, vr.grad) < 1.5e-2, rel_err(dv_got, vr.grad)
    assert rel_err(dq, qr.grad) < 2e-2, rel_err(dq, qr.grad)
    assert rel_err(dk, kr.grad) < 2e-2, rel_err(dk, kr.grad)


# ---------------------------------------------------------------------------------------- dropout (attend.py:131, :346)
def _attn_bits(L, BH, Np, p, seed, stream):
    W = L.lib().vbx_dropout_bits_words(Np)
    rm = torch.full((BH, Np, W), -1, dtype=torch.int32, device=dev)
    cm = torch.full((BH, Np, W), -1, dtype=torch.int32, device=dev)
    L.call("vbx_attn_dropout_bits", rm, cm, BH, Np, seed, stream, p, st())
    torch.cuda.synchronize()
    return rm, cm


@pytest.mark.parametrize("BH,Np,p", [(3, 100, 0.1), (2, 1040, 0.25), (1, 64, 0.5), (5, 33, 0.9)])
def test_attn_dropout_bits_are_the_documented_philox_stream(L, BH, Np, p):
    """The keep bits equal the host restatement of include/vbx.h's definition (Philox4x32-10, counter (4 * (key / 32) +
    (key % 32) / 8, q, bh, stream), 16-bit lots) in BOTH orientations; bits past Np are zero."""
    import philox_ref as PR
    seed, stream = 0x1234_5678_9ABC_DEF0 >> 2, 7
    rm, cm = _attn_bits(L, BH, Np, p, seed, stream)
    W = rm.shape[-1]
    assert W == 2 * ((Np + 63) // 64)
    want = PR.attn_keep(BH, Np, p, seed, stream)  # [bh, q, key]
    got_r = PR.unpack_bits(rm.cpu().numpy(), W * 32)
    got_c = PR.unpack_bits(cm.cpu().numpy(), W * 32)
    assert (got_r[:, :, :Np] == want).all()
    assert (got_c[:, :, :Np] == want.transpose(0, 2, 1)).all()
    assert not got_r[:, :, Np:].any() and not got_c[:, :, Np:].any()
    # a different stream / seed gives a different mask; the same arguments the same one
    rm2, _ = _attn_bits(L, BH, Np, p, seed, stream + 1)
    rm3, _ = _attn_bits(L, BH, Np, p, seed + 1, stream)
    rm4, cm4 = _attn_bits(L, BH, Np, p, seed, stream)
    assert not torch.equal(rm, rm2) and not torch.equal(rm, rm3) and torch.equal(rm, rm4) and torch.equal(cm, cm4)


def test_attn_dropout_bits_statistics(L):
    """Kept fraction = thr16 / 65536 within 5 sigma over 34 M draws; no visible dependence between neighbouring keys, neighbouring
    queries or heads; vbx_dropout_keep_scale is the exact inverse of the kept fraction."""
    import philox_ref as PR
    BH, Np, p = 32, 1040, 0.1
    rm, _ = _attn_bits(L, BH, Np, p, 99, 0)
    keep = torch.from_numpy(PR.unpack_bits(rm.cpu().numpy(), Np)).float()
    n = keep.numel()
    frac = PR.thr16(p) / 65536.0
    assert abs(L.lib().vbx_dropout_keep_scale(p) * frac - 1.0) < 1e-6
    assert abs(float(keep.mean()) - frac) < 5 * math.sqrt(frac * (1 - frac) / n)
    c = keep - frac
    for a, b in ((c[:, :, 1:], c[:, :, :-1]), (c[:, 1:], c[:, :-1]), (c[1:], c[:-1])):
        corr = float((a * b).mean()) / (frac * (1 - frac))
        assert abs(corr) < 5 / math.sqrt(a.numel()), corr


@pytest.mark.parametrize("rows,cols,ld", [(100, 64, 64), (8320, 1408, 1408), (33, 40, 48)])
def test_dropout_rows(L, rows, cols, ld):
    """nn.Dropout between GEGLU and the output projection (voicebox_pytorch.py:346): in place on the fp16 and the bf16 copy, the
    documented Philox stream, survivors scaled by 65536 / thr16."""
    import philox_ref as PR
    p, seed, stream = 0.2, 424242, 5
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, ld, generator=g) * 3
    xh, xb = x.half().to(dev), x.bfloat16().to(dev)
    L.call("vbx_dropout_rows", xh, xb, rows, cols, ld, seed, stream, p, st())
    torch.cuda.synchronize()
    keep = torch.from_numpy(PR.rows_keep(rows, cols, p, seed, stream))
    rk = L.lib().vbx_dropout_keep_scale(p)
    for got, src, tol in ((xh.float().cpu(), x.half().float(), 1e-3), (xb.float().cpu(), x.bfloat16().float(), 8e-3)):
        want = torch.where(keep, src[:, :cols] * rk, torch.zeros(()))
        assert torch.allclose(got[:, :cols], want, rtol=tol, atol=1e-6)
        assert torch.equal(got[:, cols:], src[:, cols:])  # the padding columns between cols and ld are not touched
    # one copy only
    yb = x.bfloat16().to(dev)
    L.call("vbx_dropout_rows", None, yb, rows, cols, ld, seed, stream, p, st())
    assert torch.equal(yb, xb)


@pytest.mark.parametrize("Bsz,H,Np,scale,masked,p", [(1, 2, 64, 10.0, False, 0.1), (2, 2, 1040, 10.0, False, 0.1),
                                                     (2, 3, 77, 10.0, True, 0.3), (1, 2, 200, 0.125, True, 0.5),
                                                     (2, 2, 130, 10.0, False, 0.2)])
def test_attn_dropout_fwd_bwd(L, Bsz, H, Np, scale, masked, p):
    """attend.py:121-135 with attn_dropout: softmax -> dropout -> P.V, against the fp64 restatement given the SAME keep mask (read
    back from the kernel's bits); forward statistics (LSE) are those of the undropped softmax."""
    import philox_ref as PR
    q16, k16, v = attn_inputs(Bsz, H, Np, seed=Np + H + 1, qnorm=8.0 if scale == 10.0 else None)
    mask = None
    if masked:
        mask = torch.ones(Bsz, Np, dtype=torch.bool)
        mask[0, Np - 13:] = False
        if Bsz > 1:
            mask[1, 5:9] = False
    seed = 77 + Np
    rm, cm = _attn_bits(L, Bsz * H, Np, p, seed, 4)
    keep = torch.from_numpy(PR.unpack_bits(rm.cpu().numpy(), Np)).view(Bsz, H, Np, Np)
    mult = keep.double() * L.lib().vbx_dropout_keep_scale(p)
    out16 = torch.empty(Bsz, Np, H * 64, dtype=torch.float16, device=dev)
    out = torch.empty(Bsz, Np, H * 64, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(Bsz, H, Np, device=dev)
    qs, q_eff = qpre(L, q16, scale)
    qd, kd, vd = qs.to(dev), k16.to(dev), v.to(dev)
    md = mask.to(dev) if masked else None
    L.call("vbx_attn_fwd_dropout", qd, kd, vd, md, out16, out, lse, Bsz, H, Np, scale, rm, p, st())
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q_eff, k16, v))
    ref = restate.attend(qr, kr, vr, mask=mask, scale=scale, drop=mult)
    ref_t = ref.permute(0, 2, 1, 3).reshape(Bsz, Np, H * 64)
    assert rel_err(out16, ref_t) < 1.5e-3, rel_err(out16, ref_t)
    assert rel_err(out, ref_t) < 5e-3, rel_err(out, ref_t)
    sim = torch.einsum("bhid,bhjd->bhij", qr, kr) * scale
    if masked:
        sim = sim.masked_fill(~mask[:, None, None, :], -float("inf"))
    assert max_err(lse, torch.logsumexp(sim, dim=-1) / math.log(2.0)) < 2e-3
    # the dropped output differs from the undropped one by far more than the tolerance above (the mask is really applied)
    plain = restate.attend(qr, kr, vr, mask=mask, scale=scale).permute(0, 2, 1, 3).reshape(Bsz, Np, H * 64)
    assert rel_err(out16, plain) > 0.05
    # ---- backward
    g = torch.Generator().manual_seed(5)
    dout = bf(torch.randn(Bsz, Np, H * 64, generator=g) * 1e-3)
    ref_t.backward(dout.double())
    qb, kb = bf(q16.float()).to(dev), bf(k16.float()).to(dev)
    delta = torch.empty(Bsz, H, Np, device=dev)
    dq = torch.zeros(Bsz, H, Np, 64, device=dev)
    dk = torch.zeros(Bsz, H, Np, 64, device=dev)
    dv = torch.zeros(Bsz, Np, 3 * H * 64, dtype=torch.bfloat16, device=dev)
    dv_view = dv[:, :, 2 * H * 64:]
    L.call("vbx_attn_bwd_dropout", qd, kd, qb, kb, bf(v.float()).to(dev), md, out16, 1, dout.to(dev), lse, delta, dq, dk,
           dv_view.data_ptr(), 3 * H * 64, Bsz, H, Np, scale, rm, cm, p, st())
    torch.cuda.synchronize()
    dv_got = dv_view.float().cpu().view(Bsz, Np, H, 64).permute(0, 2, 1, 3)
    assert rel_err(dv_got, vr.grad) < 1.5e-2, rel_err(dv_got, vr.grad)
    # At |q| = |k| = 8, scale 10 the softmax is one-hot: where the top key survives, dS = P (dP / keep - delta) cancels to ~0 and what
    # remains is the fp16 rounding of O inside delta = dO . O -- a CPU emulation of exactly these roundings (fp16 P and O, bf16 dO, V,
    # dS, q, k; fp64 otherwise) gives 3.4 % / 2.2 % / 1.4 % for Np = 64 / 77 / 130 and the kernels land on those figures (3.6 / 2.3 %).
    tol = 5e-2 if scale == 10.0 else 2e-2
    assert rel_err(dq, qr.grad) < tol, rel_err(dq, qr.grad)
    assert rel_err(dk, kr.grad) < tol, rel_err(dk, kr.grad)


@pytest.mark.parametrize("qknorm", [True, False])
def test_qknorm_rope_bwd(L, qknorm):
    Bsz, H, Np = 2, 2, 70
    g = torch.Generator().manual_seed(9)
    t = torch.randn(2, Bsz, H, Np, 64, generator=g, dtype=torch.float64, requires_grad=True)
    gam = (1 + 0.2 * torch.randn(2, H, 64, generator=g, dtype=torch.float64)).requires_grad_(True)
    fr, rc, rs = rot_tables(Np, 16)
    outs = []
    for w in range(2):
        y = restate.l2norm_scale(t[w], 64) * gam[w][:, None, :] if qknorm else t[w]
        outs.append(restate.apply_rotary(fr.double(), y))
    up = torch.randn(2, Bsz, H, Np, 64, generator=g)
    (outs[0] * up[0].double()).sum().backward(retain_graph=True)
    (outs[1] * up[1].double()).sum().backward()
    rn = (1 / t.detach().norm(dim=-1)).float()
    dqkv = torch.zeros(Bsz * Np, 3 * H * 64, dtype=torch.bfloat16, device=dev)
    rows = L.lib().vbx_qknorm_rope_bwd_gpart_rows(Bsz)
    gpart = torch.zeros(2, rows, H, 64, device=dev)
    L.call("vbx_qknorm_rope_bwd", up[0].to(dev), up[1].to(dev), outs[0].detach().half().to(dev),
           outs[1].detach().half().to(dev), rn[0].to(dev), rn[1].to(dev), gam[0].detach().float().to(dev),
           gam[1].detach().float().to(dev), rc.to(dev), rs.to(dev), 8.0 if qknorm else 0.0, dqkv, 3 * H * 64, gpart,
           Bsz, H, Np, 1.0, st())  # q16 passed unscaled here: q16_scale = 1
    got = dqkv.float().cpu().view(Bsz, Np, 3, H, 64).permute(2, 0, 3, 1, 4)
    for w in range(2):
        assert rel_err(got[w], t.grad[w]) < 5e-3, (w, rel_err(got[w], t.grad[w]))
    if qknorm:
        assert rel_err(gpart.sum(1), gam.grad) < 2e-3


@pytest.mark.parametrize("Bsz,H,Np,qknorm,masked", [(2, 2, 1040, True, False), (2, 2, 77, True, True), (1, 2, 200, False, False)])
def test_attn_bwd_fused_equals_two_pass(L, Bsz, H, Np, qknorm, masked, bwd_variant):
    """vbx_attn_bwd_fused (rotary + qk-norm backward inside the dq / dkdv epilogues) against vbx_attn_bwd followed by
    vbx_qknorm_rope_bwd on the same inputs: same d(qkv) up to the bf16 rounding of the output, same gamma gradients."""
    scale = 10.0 if qknorm else 0.125
    g = torch.Generator().manual_seed(Np)
    pre = torch.randn(2, Bsz, H, Np, 64, generator=g)
    gam = 1 + 0.2 * torch.randn(2, H, 64, generator=g)
    fr, rc, rs = rot_tables(Np, 16 if Np > 16 else 0)
    hats = []
    for w in range(2):
        y = restate.l2norm_scale(pre[w], 64) * gam[w][:, None, :] if qknorm else pre[w]
        hats.append(restate.apply_rotary(fr, y))
    rn = (1 / pre.norm(dim=-1)).float()
    q16, k16 = qpre(L, hats[0].half(), scale)[0].to(dev), hats[1].half().to(dev)  # q16 in the kernels' exp2 domain
    v = torch.randn(Bsz, H, Np, 64, generator=g).half().to(dev)
    mask = None
    if masked:
        mask = torch.ones(Bsz, Np, dtype=torch.bool)
        mask[0, Np - 9:] = False
        mask = mask.to(dev)
    out16 = torch.empty(Bsz, Np, H * 64, dtype=torch.float16, device=dev)
    lse = torch.empty(Bsz, H, Np, device=dev)
    L.call("vbx_attn_fwd", q16, k16, v, mask, out16, None, lse, Bsz, H, Np, scale, st())
    dout = bf(torch.randn(Bsz, Np, H * 64, generator=g) * 1e-3).to(dev)
    qb, kb, vb = bf(hats[0].half().float()).to(dev), bf(k16.float()), bf(v.float())
    I = H * 64
    # two-pass
    delta = torch.empty(Bsz, H, Np, device=dev)
    dq, dk = torch.zeros(Bsz, H, Np, 64, device=dev), torch.zeros(Bsz, H, Np, 64, device=dev)
    d1 = torch.zeros(Bsz * Np, 3 * I, dtype=torch.bfloat16, device=dev)
    L.call("vbx_attn_bwd", q16, k16, qb, kb, vb, mask, out16, 1, dout, lse, delta, dq, dk, d1.view(-1)[2 * I:].data_ptr(), 3 * I,
           Bsz, H, Np, scale, attn_scratch(L, Bsz, H, Np), st())
    rows1 = L.lib().vbx_qknorm_rope_bwd_gpart_rows(Bsz)
    gp1 = torch.zeros(2, rows1, H, 64, device=dev)
    gq, gk = gam[0].float().to(dev), gam[1].float().to(dev)
    L.call("vbx_qknorm_rope_bwd", dq, dk, q16, k16, rn[0].to(dev), rn[1].to(dev), gq, gk, rc.to(dev), rs.to(dev),
           8.0 if qknorm else 0.0, d1, 3 * I, gp1, Bsz, H, Np, L.lib().vbx_attn_q_prescale(scale), st())
    # fused
    d2 = torch.zeros(Bsz * Np, 3 * I, dtype=torch.bfloat16, device=dev)
    rows2 = Bsz * L.lib().vbx_attn_bwd_fused_tiles(Np)
    gp2 = torch.zeros(2, rows2, H, 64, device=dev)
    L.call("vbx_attn_bwd_fused", q16, k16, qb, kb, vb, mask, out16, 1, dout, lse, delta, rn[0].to(dev), rn[1].to(dev), gq, gk,
           rc.to(dev), rs.to(dev), 8.0 if qknorm else 0.0, d2, 3 * I, gp2, Bsz, H, Np, scale, attn_scratch(L, Bsz, H, Np), st())
    torch.cuda.synchronize()
    assert torch.equal(d1[:, 2 * I:], d2[:, 2 * I:])  # dv: same code path
    for blk in range(2):
        a, b = d1[:, blk * I:(blk + 1) * I].float(), d2[:, blk * I:(blk + 1) * I].float()
        assert rel_err(b, a) < 4e-3, (blk, rel_err(b, a))  # identical fp32 math, one bf16 rounding each
    if qknorm:
        assert rel_err(gp2.sum(1), gp1.sum(1)) < 1e-4


def _bwd_case(L, Bsz, H, Np, seed, masked=False):
    """Inputs of one fused attention backward at the reference's logit scale (qk-norm, rotary, scale 10)."""
    g = torch.Generator().manual_seed(seed)
    pre = torch.randn(2, Bsz, H, Np, 64, generator=g)
    gam = 1 + 0.2 * torch.randn(2, H, 64, generator=g)
    fr, rc, rs = rot_tables(Np, 16 if Np > 16 else 0)
    hats = [restate.apply_rotary(fr, restate.l2norm_scale(pre[w], 64) * gam[w][:, None, :]) for w in range(2)]
    rn = (1 / pre.norm(dim=-1)).float()
    c = dict(Bsz=Bsz, H=H, Np=Np, q16=qpre(L, hats[0].half(), 10.0)[0].to(dev), qb=bf(hats[0].half().float()).to(dev),
             k16=hats[1].half().to(dev),
             v=torch.randn(Bsz, H, Np, 64, generator=g).half().to(dev), rn=rn.to(dev), gam=gam.float().to(dev), rc=rc.to(dev),
             rs=rs.to(dev), mask=None)
    if masked:
        m = torch.ones(Bsz, Np, dtype=torch.bool)
        m[0, Np - 9:] = False
        m[Bsz - 1, 3:7] = False
        c["mask"] = m.to(dev)
    c["out16"] = torch.empty(Bsz, Np, H * 64, dtype=torch.float16, device=dev)
    c["lse"] = torch.empty(Bsz, H, Np, device=dev)
    L.call("vbx_attn_fwd", c["q16"], c["k16"], c["v"], c["mask"], c["out16"], None, c["lse"], Bsz, H, Np, 10.0, st())
    c["dout"] = bf(torch.randn(Bsz, Np, H * 64, generator=g) * 1e-3).to(dev)
    return c


def _bwd_fused(L, c, variant, scratch=None):
    Bsz, H, Np = c["Bsz"], c["H"], c["Np"]
    I = H * 64
    L.lib().vbx_attn_bwd_select(variant)
    try:
        d = torch.zeros(Bsz * Np, 3 * I, dtype=torch.bfloat16, device=dev)
        gp = torch.zeros(2, Bsz * L.lib().vbx_attn_bwd_fused_tiles(Np), H, 64, device=dev)
        delta = torch.empty(Bsz, H, Np, device=dev)
        scratch = attn_scratch(L, Bsz, H, Np) if scratch is None else scratch
        L.call("vbx_attn_bwd_fused", c["q16"], c["k16"], c["qb"], bf(c["k16"].float()), bf(c["v"].float()), c["mask"],
               c["out16"], 1, c["dout"], c["lse"], delta, c["rn"][0], c["rn"][1], c["gam"][0], c["gam"][1], c["rc"], c["rs"], 8.0, d,
               3 * I, gp, Bsz, H, Np, 10.0, scratch, st())
        torch.cuda.synchronize()
    finally:
        L.lib().vbx_attn_bwd_select(0)
    return d, gp, scratch


@pytest.mark.parametrize("Bsz,H,Np,masked", [(8, 16, 1040, False), (2, 4, 1040, True), (3, 5, 520, False), (1, 2, 130, True), (1, 2, 24, False)])
def test_attn_bwd_folded_statistics_equal_the_unfolded_bodies(L, Bsz, H, Np, masked):
    """Round 5's default backward (csrc/attn_bwd_fold.inc: L and delta enter as the C operand of the S / dP MFMA chains, negated
    stationary fragments, P = exp2(-(L - q.k))) against round 3's bodies (select 3: P = exp2(fma(s, 1, -L)), dP - delta by VALU) on the
    same inputs, incl. the benchmark grid.  The two differ only in where the fp32 rounding of the exponent / of dP - delta happens, then
    share the bf16 rounding of P and dS: every output within bf16 noise of the other, gamma partials likewise, and the folded kernel is
    deterministic."""
    c = _bwd_case(L, Bsz, H, Np, seed=7 * Np + Bsz, masked=masked)
    d3, g3, _ = _bwd_fused(L, c, 3)
    d1, g1, _ = _bwd_fused(L, c, 1)
    d1b, g1b, _ = _bwd_fused(L, c, 1)
    assert torch.equal(d1, d1b) and torch.equal(g1, g1b)
    I = H * 64
    assert torch.isfinite(d1.float()).all()
    for blk, name in ((0, "dq"), (1, "dk"), (2, "dv")):
        e = rel_err(d1[:, blk * I:(blk + 1) * I].float(), d3[:, blk * I:(blk + 1) * I].float())
        assert e < 6e-3, (name, e)
    assert rel_err(g1.sum(1), g3.sum(1)) < 2e-3


# ----------------------------------------------------------------------------- small ops
@pytest.mark.parametrize("masked,ks", [(False, 31), (True, 31), (True, 7), (False, 1), (True, 17), (False, 29)])
def test_convpos_fwd_bwd(L, masked, ks):
    """ConvPositionEmbed (voicebox_pytorch.py:203-233) at the reference's default kernel size and at other odd sizes."""
    Bsz, N, R, D = 2, 150, 16, 128
    g = torch.Generator().manual_seed(2)
    e = torch.randn(Bsz, N, D, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(D, 1, ks, generator=g, dtype=torch.float64) * ks ** -0.5).requires_grad_(True)
    b = (0.1 * torch.randn(D, generator=g, dtype=torch.float64)).requires_grad_(True)
    reg = torch.randn(R, D, generator=g)
    mask = None
    if masked:
        mask = torch.ones(Bsz, N, dtype=torch.bool)
        mask[1, 100:] = False
    ref = restate.conv_pos_embed(e, w, b, mask) + e
    xs = torch.zeros(Bsz, N + R, D, device=dev)
    ed, wd, bd = e.detach().float().to(dev), w.detach().float().to(dev), b.detach().float().to(dev)
    md = mask.to(dev) if masked else None
    L.call("vbx_convpos_fwd", ed, wd, bd, md, reg.to(dev), xs, Bsz, N, R, D, ks, st())
    assert rel_err(xs[:, R:], ref) < 1e-5
    assert torch.equal(xs[:, :R].cpu(), reg.expand(Bsz, R, D))
    dxs = torch.randn(Bsz, N + R, D, generator=g)
    ref.backward(dxs[:, R:].double())
    chunks = L.lib().vbx_convpos_bwd_chunks(Bsz, N)
    dpre = torch.empty(Bsz, N, D, device=dev)
    de = torch.empty(Bsz, N, D, device=dev)
    deb = torch.empty(Bsz, N, D, dtype=torch.bfloat16, device=dev)
    wpart = torch.zeros(chunks, D, 64, device=dev)
    dreg = torch.zeros(R, D, device=dev)
    L.call("vbx_convpos_bwd", ed, wd, bd, md, dxs.to(dev), dpre, de, deb, wpart, dreg, Bsz, N, R, D, ks, st())
    assert rel_err(de, e.grad) < 1e-5
    assert rel_err(deb, e.grad) < 4e-3
    dw, db = torch.zeros(D, 1, ks, device=dev), torch.zeros(D, device=dev)
    L.call("vbx_conv_wgrad_finalize", wpart, chunks, D, ks, dw, db, st())
    assert rel_err(dw, w.grad) < 1e-5
    assert rel_err(db, b.grad) < 1e-5
    assert rel_err(dreg, dxs[:, :R].sum(0)) < 1e-6


def test_time_embed_and_adaln(L):
    Bsz, D, Th, J = 3, 64, 256, 512
    g = torch.Generator().manual_seed(4)
    times = torch.rand(Bsz, generator=g)
    p = {"sinu_pos_emb.0.weights": torch.randn(D // 2, generator=g),
         "sinu_pos_emb.1.weight": torch.randn(Th, D, generator=g) * D ** -0.5,
         "sinu_pos_emb.1.bias": torch.randn(Th, generator=g) * 0.1}
    pr = {k: v.double().requires_grad_(True) for k, v in p.items()}
    temb_ref = restate.time_embedding(times.double(), pr)
    four = torch.empty(Bsz, D, device=dev)
    pre = torch.empty(Bsz, Th, device=dev)
    temb = torch.empty(Bsz, Th, device=dev)
    pd = {k: v.to(dev) for k, v in p.items()}
    L.call("vbx_time_embed_fwd", times.to(dev), pd["sinu_pos_emb.0.weights"], pd["sinu_pos_emb.1.weight"],
           pd["sinu_pos_emb.1.bias"], four, pre, temb, Bsz, D, Th, st())
    assert rel_err(temb, temb_ref) < 1e-5
    # adaLN projection
    W = torch.randn(J, Th, generator=g) * 0.02
    bias = torch.randn(J, generator=g)
    Wb = W.half().to(dev)
    Wr = W.half().double().requires_grad_(True)
    br = bias.double().requires_grad_(True)
    ada_ref = temb_ref @ Wr.t() + br
    ada = torch.empty(Bsz, J, device=dev)
    L.call("vbx_adaln_proj_fwd", temb, Wb, bias.to(dev), ada, Bsz, Th, J, 0, st())
    assert rel_err(ada, ada_ref) < 1e-5
    ada_g = torch.empty(J // 128, Bsz, 128, device=dev)  # grouped layout used by the runtime: [layer][b][4D]
    L.call("vbx_adaln_proj_fwd", temb, Wb, bias.to(dev), ada_g, Bsz, Th, J, 128, st())
    assert torch.equal(ada_g.permute(1, 0, 2).reshape(Bsz, J), ada)
    dada = torch.randn(Bsz, J, generator=g)
    ada_ref.backward(dada.double())
    dW = torch.empty(J, Th, device=dev)
    dbias = torch.empty(J, device=dev)
    dtemb = torch.empty(Bsz, Th, device=dev)
    scratch = torch.empty(L.lib().vbx_adaln_proj_bwd_scratch_floats(Bsz, Th, J), device=dev)
    L.call("vbx_adaln_proj_bwd", temb, Wb, dada.to(dev), dW, dbias, dtemb, scratch, Bsz, Th, J, 0, st())
    assert rel_err(dW, Wr.grad) < 1e-5 and rel_err(dbias, br.grad) < 1e-5
    # time-embedding backward from the oracle's d(temb)
    dtemb_ref = dada.double() @ Wr.detach()
    assert rel_err(dtemb, dtemb_ref) < 1e-5
    dws = torch.empty(D // 2, device=dev)
    dw1 = torch.empty(Th, D, device=dev)
    db1 = torch.empty(Th, device=dev)
    sc = torch.empty(L.lib().vbx_time_embed_bwd_scratch_floats(Bsz, D), device=dev)
    L.call("vbx_time_embed_bwd", times.to(dev), pd["sinu_pos_emb.0.weights"], pd["sinu_pos_emb.1.weight"], four, pre,
           dtemb, dws, dw1, db1, sc, Bsz, D, Th, st())
    assert rel_err(dw1, pr["sinu_pos_emb.1.weight"].grad) < 1e-4
    assert rel_err(db1, pr["sinu_pos_emb.1.bias"].grad) < 1e-4
    assert rel_err(dws, pr["sinu_pos_emb.0.weights"].grad) < 1e-4


@pytest.mark.parametrize("Bsz,Th,J", [(8, 2048, 2048), (11, 264, 516), (2, 64, 8192), (1, 8, 4)])
def test_adaln_proj_bwd_shapes(L, Bsz, Th, J):
    """dW = dada^T temb, dbias, dtemb = dada W at the bench shape, with a batch > 8 (two LDS rounds), a J that is not a multiple
    of the slice count, the largest J of the LDS-staged kernel (dim 2048) and a degenerate shape."""
    g = torch.Generator().manual_seed(Bsz + Th + J)
    temb = torch.randn(Bsz, Th, generator=g)
    W = (torch.randn(J, Th, generator=g) * 0.02).half()
    dada = torch.randn(Bsz, J, generator=g)
    dW = torch.empty(J, Th, device=dev)
    dbias = torch.empty(J, device=dev)
    dtemb = torch.empty(Bsz, Th, device=dev)
    scratch = torch.full((L.lib().vbx_adaln_proj_bwd_scratch_floats(Bsz, Th, J),), float("nan"), device=dev)
    L.call("vbx_adaln_proj_bwd", temb.to(dev), W.to(dev), dada.to(dev), dW, dbias, dtemb, scratch, Bsz, Th, J, 0, st())
    assert rel_err(dW, dada.double().t() @ temb.double()) < 1e-5
    assert rel_err(dbias, dada.double().sum(0)) < 1e-5
    assert rel_err(dtemb, dada.double() @ W.double()) < 1e-5
    keep = dtemb.clone()
    L.call("vbx_adaln_proj_bwd", temb.to(dev), W.to(dev), dada.to(dev), dW, dbias, dtemb, scratch, Bsz, Th, J, 1, st())
    assert rel_err(dtemb, 2 * keep) < 1e-6  # accumulate_dtemb


@pytest.mark.parametrize("Lyr,Bsz,Th,J", [(12, 8, 2048, 2048), (3, 11, 264, 516), (2, 1, 8, 4)])
def test_adaln_dtemb_all_layers(L, Lyr, Bsz, Th, J):
    """Factor form of the adaLN weight gradient: what is left of the projections' backward is d(time_emb) = sum over layers of
    dada_l W_l, one launch for all layers (vbx_adaln_dtemb_all) -- against fp64, and against the per-layer entry point accumulated."""
    g = torch.Generator().manual_seed(Lyr + Bsz + Th + J)
    W = (torch.randn(Lyr, J, Th, generator=g) * 0.02).half()
    dada = torch.randn(Lyr, Bsz, J, generator=g)
    temb = torch.randn(Bsz, Th, generator=g).to(dev)
    scratch = torch.full((L.lib().vbx_adaln_dtemb_all_scratch_floats(Lyr, Bsz, Th, J),), float("nan"), device=dev)
    dtemb = torch.full((Bsz, Th), float("nan"), device=dev)
    L.call("vbx_adaln_dtemb_all", W.to(dev), dada.to(dev), dtemb, scratch, Lyr, Bsz, Th, J, st())
    want = torch.einsum("lbj,ljt->bt", dada.double(), W.double())
    assert rel_err(dtemb, want) < 1e-5
    per = torch.empty(Bsz, Th, device=dev)
    dbias = torch.empty(J, device=dev)
    sc = torch.empty(L.lib().vbx_adaln_proj_bwd_scratch_floats(Bsz, Th, J), device=dev)
    for l in range(Lyr):
        L.call("vbx_adaln_proj_bwd", temb, W[l].contiguous().to(dev), dada[l].contiguous().to(dev), None, dbias, per, sc, Bsz, Th, J, int(l > 0), st())
    assert rel_err(dtemb, per) < 1e-5


def test_layer_reduce_equals_the_two_launches(L):
    """vbx_layer_reduce = vbx_splitk_reduce_multi + vbx_multi_reduce in ONE launch: bit-identical outputs (weight-gradient slabs with
    and without the GEGLU row un-interleave, ragged sizes; a plain and a batched column reduction)."""
    import ctypes as C

    class SJob(C.Structure):
        _fields_ = [("slabs", C.c_void_p), ("dst", C.c_void_p), ("sq", C.c_void_p)] + [(n, C.c_int) for n in
                    ("splits", "M", "N", "dst_rows", "dst_cols", "dst_ld", "rowmap", "F", "block0", "pad_")]

    class SJobs(C.Structure):
        _fields_ = [("job", SJob * 6), ("n", C.c_int)]  # VBX_SKR_MAX

    class MJob(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_bstride", C.c_long), ("dst_bstride", C.c_long),
                    ("row_stride", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("batches", C.c_int), ("dst_len", C.c_int),
                    ("rowmap", C.c_int), ("F", C.c_int), ("block0", C.c_int), ("pad_", C.c_int)]

    class MJobs(C.Structure):
        _fields_ = [("job", MJob * 48), ("n", C.c_int)]  # VBX_MR_MAX

    g = torch.Generator().manual_seed(31)
    Fd, Fp = 170, 192
    specs = [(5, 3 * 128, 512, 3 * 128, 512, 0, 0), (3, 2 * Fp, 264, 2 * Fd, 264, 1, Fd), (7, 72, 1028, 72, 1026, 0, 0)]
    slabs = [torch.randn(sp, M, N, generator=g).to(dev) for sp, M, N, *_ in specs]
    x0 = torch.randn(130, 300, generator=g).to(dev)
    x1 = torch.randn(3, 37, 100, generator=g).to(dev)

    lib0 = L.lib()
    lib0.vbx_splitk_reduce_blocks.argtypes = [C.c_int, C.c_int]
    sqs = [torch.full((lib0.vbx_splitk_reduce_blocks(M, N),), float("nan"), device=dev) for _, M, N, *_ in specs]

    def run(fused):
        outs = [torch.full((dr, dc), 7.0, device=dev) for _, _, _, dr, dc, _, _ in specs]
        o0, o1 = torch.full((300,), 7.0, device=dev), torch.full((3, 80), 7.0, device=dev)
        sj, mj = SJobs(), MJobs()
        sj.n, mj.n = len(specs), 2
        for i, (sp, M, N, dr, dc, rm, F) in enumerate(specs):
            j = sj.job[i]
            j.slabs, j.dst, j.splits, j.M, j.N, j.dst_rows, j.dst_cols, j.dst_ld, j.rowmap, j.F = slabs[i].data_ptr(), outs[i].data_ptr(), sp, M, N, dr, dc, dc, rm, F
            j.sq = None if fused else sqs[i].data_ptr()  # vbx_skr_job.sq: per-block sums of squares of what was stored (multi launch only)
        a, b = mj.job[0], mj.job[1]
        a.src, a.dst, a.rows, a.cols, a.row_stride, a.batches, a.dst_len = x0.data_ptr(), o0.data_ptr(), 130, 300, 300, 1, 300
        b.src, b.dst, b.rows, b.cols, b.row_stride, b.batches = x1.data_ptr(), o1.data_ptr(), 37, 64, 100, 3
        b.src_bstride, b.dst_bstride, b.dst_len = 37 * 100, 80, 64
        lib = L.lib()
        lib.vbx_layer_reduce.argtypes = [C.POINTER(SJobs), C.POINTER(MJobs), C.c_void_p]
        lib.vbx_splitk_reduce_multi.argtypes = [C.POINTER(SJobs), C.c_void_p]
        lib.vbx_multi_reduce.argtypes = [C.POINTER(MJobs), C.c_void_p]
        if fused:
            assert lib.vbx_layer_reduce(C.byref(sj), C.byref(mj), st()) == 0, lib.vbx_last_error()
        else:
            assert lib.vbx_splitk_reduce_multi(C.byref(sj), st()) == 0, lib.vbx_last_error()
            assert lib.vbx_multi_reduce(C.byref(mj), st()) == 0, lib.vbx_last_error()
        torch.cuda.synchronize()
        return outs + [o0, o1]

    sep, fus = run(False), run(True)
    for a, b in zip(sep, fus):
        assert torch.equal(a, b)
    assert rel_err(fus[0], slabs[0].double().sum(0)) < 1e-6 and rel_err(fus[3], x0.double().sum(0)) < 1e-6
    assert rel_err(fus[2], slabs[2].double().sum(0)[:, :1026]) < 1e-6  # ragged: unaligned row stride, columns past dst_cols dropped
    # the gradient-norm partials: sum over blocks = sum of squares of exactly the stored values (dropped GEGLU rows / columns excluded),
    # and the same bits on a second run
    first = [q.clone() for q in sqs]
    run(False)
    for i, q in enumerate(sqs):
        assert torch.equal(q, first[i])
        want = float(sep[i].double().pow(2).sum())
        assert abs(float(q.double().sum()) - want) < 1e-6 * want, (i, float(q.double().sum()), want)


def test_geglu_bwd_and_colsum(L):
    M, Fd, Fp = 100, 170, 192
    g = torch.Generator().manual_seed(6)
    h1 = bf(torch.randn(M, 2 * Fp, generator=g))
    dg = bf(torch.randn(M, Fp, generator=g))
    dh1 = torch.empty(M, 2 * Fp, dtype=torch.bfloat16, device=dev)
    L.call("vbx_geglu_bwd", h1.to(dev), dg.to(dev), dh1, M, Fp, st())
    blk = h1.double().view(M, Fp // 64, 2, 64)
    xr = blk[:, :, 0].reshape(M, Fp).requires_grad_(True)
    gr = blk[:, :, 1].reshape(M, Fp).requires_grad_(True)
    (F.gelu(gr) * xr).backward(dg.double())
    got = dh1.float().cpu().view(M, Fp // 64, 2, 64)
    assert rel_err(got[:, :, 0].reshape(M, Fp), xr.grad) < 4e-3
    assert rel_err(got[:, :, 1].reshape(M, Fp), gr.grad) < 4e-3
    # column sums with the GEGLU un-mapping (bias grad of FeedForward[0])
    out = torch.zeros(2 * Fd, device=dev)
    scratch = torch.empty(L.lib().vbx_colsum_scratch_floats(M, 2 * Fp), device=dev)
    L.call("vbx_colsum_bf16", dh1, M, 2 * Fp, 2 * Fp, out, 2 * Fd, 1, Fd, scratch, st())
    cs = dh1.float().cpu().double().sum(0).view(Fp // 64, 2, 64)
    exp = torch.cat((cs[:, 0].reshape(-1)[:Fd], cs[:, 1].reshape(-1)[:Fd]))
    assert rel_err(out, exp) < 1e-5
    x32 = torch.randn(M, 96, generator=g)
    big = bf(torch.randn(8320, 2816, generator=g))  # the FeedForward bias-grad shape of the benchmark config
    ob = torch.zeros(2816, device=dev)
    scb = torch.empty(L.lib().vbx_colsum_scratch_floats(8320, 2816), device=dev)
    L.call("vbx_colsum_bf16", big.to(dev), 8320, 2816, 2816, ob, 2816, 0, 0, scb, st())
    assert rel_err(ob, big.double().sum(0)) < 1e-5
    o2 = torch.zeros(96, device=dev)
    sc2 = torch.empty(L.lib().vbx_colsum_scratch_floats(M, 96), device=dev)
    L.call("vbx_colsum_f32", x32.to(dev), M, 96, 96, o2, sc2, st())
    assert rel_err(o2, x32.double().sum(0)) < 1e-6


def test_masked_mse_cfm_axpy(L):
    Bsz, N, D = 3, 50, 64
    g = torch.Generator().manual_seed(8)
    pred = torch.randn(Bsz, N, D, generator=g, dtype=torch.float64, requires_grad=True)
    target = torch.randn(Bsz, N, D, generator=g, dtype=torch.float64)
    lm = torch.rand(Bsz, N, generator=g) < 0.6
    lm[2] = False  # a sample with an empty loss mask: den clamps to 1e-5 (voicebox_pytorch.py:1112)
    per = ((pred - target) ** 2).mean(-1).masked_fill(~lm, 0.0)
    ref = (per.sum(-1) / lm.sum(-1).clamp(min=1e-5)).mean()
    ref.backward()
    per_b = torch.zeros(L.lib().vbx_masked_mse_scratch_floats(Bsz), device=dev)
    loss = torch.zeros(1, device=dev)
    pd, td, ld = pred.detach().float().to(dev), target.float().to(dev), lm.to(dev)
    L.call("vbx_masked_mse_fwd", pd, td, ld, per_b, loss, Bsz, N, D, st())
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    dpred = torch.empty(Bsz, N, D, device=dev)
    dpb = torch.empty(Bsz, N, D, dtype=torch.bfloat16, device=dev)
    L.call("vbx_masked_mse_bwd", pd, td, ld, per_b, None, dpred, dpb, Bsz, N, D, st())
    assert rel_err(dpred, pred.grad) < 1e-5
    assert rel_err(dpb, pred.grad) < 4e-3
    # CFM inputs (voicebox_pytorch.py:1404-1410)
    x1, x0 = torch.randn(Bsz, N, D, generator=g), torch.randn(Bsz, N, D, generator=g)
    times = torch.rand(Bsz, generator=g)
    for sigma in (0.0, 0.1):
        w_ref, f_ref = restate.cfm_inputs(x1, x0, times, sigma)
        w, fl = torch.empty(Bsz, N, D, device=dev), torch.empty(Bsz, N, D, device=dev)
        L.call("vbx_cfm_inputs", x1.to(dev), x0.to(dev), times.to(dev), sigma, w, fl, Bsz, N * D, st())
        assert max_err(w, w_ref) < 1e-6 and max_err(fl, f_ref) < 1e-6
    coef = torch.tensor([0.5, -0.25], device=dev)
    out = torch.empty(Bsz, N, D, device=dev)
    L.call("vbx_axpy_dev", x1.to(dev), x0.to(dev), coef, 1, out, Bsz * N * D, st())
    assert max_err(out, x1 - 0.25 * x0) < 1e-6


def test_adam_sumsq_clip(L):
    n = 100003
    g = torch.Generator().manual_seed(10)
    p = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref_p], lr=3e-4, betas=(0.9, 0.99), eps=1e-8)
    pd, m, v = p.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    ss, coef, scratch = torch.zeros(1, device=dev), torch.zeros(2, device=dev), torch.zeros(1024, device=dev)
    for step, gr in enumerate(grads, 1):
        ref_p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
        opt.step()
        gd = gr.to(dev)
        L.call("vbx_sumsq", gd, n, ss, scratch, st())
        assert abs(float(ss) - float(gr.double().pow(2).sum())) < 1e-4 * float(ss)
        L.call("vbx_clip_coef", ss, 0.5, 1.0, coef, st())
        L.call("vbx_adam_step", pd, gd, m, v, n, 3e-4, 0.9, 0.99, 1e-8, step, coef, st())
    assert max_err(pd, ref_p.detach()) < 2e-6


# ----------------------------------------------------------------------------- GateLoop (scan + post LayerNorm)
def _scan_ref(qkva):
    """oracle/restate.py:gateloop recurrence on an fp64 [B, Np, 3D] projection (autograd-able)."""
    q, kv, a = qkva.chunk(3, dim=-1)
    a = a.sigmoid()
    h = torch.zeros_like(kv[:, 0])
    hs, outs = [], []
    for t in range(qkva.shape[1]):
        h = a[:, t] * h + kv[:, t]
        hs.append(h)
        outs.append(q[:, t] * h)
    return torch.stack(outs, 1), torch.stack(hs, 1)


@pytest.mark.parametrize("Bsz,Np,D", [(2, 56, 64), (3, 1040, 96), (1, 7, 32), (2, 33, 40)])
def test_gateloop_scan_fwd_bwd(L, Bsz, Np, D):
    g = torch.Generator().manual_seed(Np + D)
    qkva = torch.randn(Bsz, Np, 3 * D, generator=g)
    qkva[..., 2 * D:] = qkva[..., 2 * D:] * 1.5 + 1.0  # gates mostly open: long memory, exercises the chunk carries
    ds = torch.randn(Bsz, Np, D, generator=g)
    s = torch.empty(Bsz, Np, D, device=dev)
    h = torch.empty(Bsz, Np, D, device=dev)
    L.call("vbx_gateloop_scan_fwd", qkva.to(dev), s, h, Bsz, Np, D, st())
    ref_in = qkva.double().requires_grad_(True)
    s_ref, h_ref = _scan_ref(ref_in)
    assert rel_err(s, s_ref) < 2e-6, rel_err(s, s_ref)
    assert rel_err(h, h_ref) < 2e-6
    s2 = torch.empty_like(s)
    L.call("vbx_gateloop_scan_fwd", qkva.to(dev), s2, None, Bsz, Np, D, st())  # eval form: no state kept
    assert torch.equal(s, s2)
    dq = torch.empty(Bsz, Np, 3 * D, dtype=torch.bfloat16, device=dev)
    L.call("vbx_gateloop_scan_bwd", qkva.to(dev), h, ds.to(dev), dq, Bsz, Np, D, st())
    s_ref.backward(ds.double())
    # output is bf16 (a GEMM operand): 2^-9 relative per element
    assert rel_err(dq.float(), ref_in.grad) < 4e-3, rel_err(dq.float(), ref_in.grad)
    for part in range(3):
        sl = slice(part * D, (part + 1) * D)
        assert rel_err(dq.float()[..., sl], ref_in.grad[..., sl]) < 4e-3, part


@pytest.mark.parametrize("Bsz,Np,D", [(2, 56, 64), (8, 1040, 512), (1, 5, 2048)])
def test_layernorm_fwd_bwd(L, Bsz, Np, D):
    g = torch.Generator().manual_seed(D)
    s = torch.randn(Bsz, Np, D, generator=g) * 2 + 0.3
    w = 1 + 0.1 * torch.randn(D, generator=g)
    b = 0.1 * torch.randn(D, generator=g)
    resid = torch.randn(Bsz, Np, D, generator=g)
    dy = torch.randn(Bsz, Np, D, generator=g)
    y = torch.empty(Bsz, Np, D, device=dev)
    L.call("vbx_layernorm_fwd", s.to(dev), w.to(dev), b.to(dev), resid.to(dev), y, Bsz * Np, D, 1e-5, st())
    sd, wd, bd = s.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(sd, (D,), wd, bd, eps=1e-5)
    assert rel_err(y, ref + resid.double()) < 2e-6
    y0 = torch.empty_like(y)
    L.call("vbx_layernorm_fwd", s.to(dev), w.to(dev), b.to(dev), None, y0, Bsz * Np, D, 1e-5, st())
    assert rel_err(y0, ref) < 2e-6
    ref.backward(dy.double())
    chunks = (Np + 15) // 16
    dsg = torch.empty(Bsz, Np, D, device=dev)
    part = torch.empty(Bsz, chunks, 2, D, device=dev)
    L.call("vbx_layernorm_bwd", s.to(dev), w.to(dev), dy.to(dev), dsg, part, Bsz, Np, D, 1e-5, st())
    assert rel_err(dsg, sd.grad) < 1e-5, rel_err(dsg, sd.grad)
    tmp = torch.empty(Bsz, 2 * D, device=dev)
    L.call("vbx_reduce_norm_partials", part, tmp, 2 * D, Bsz, chunks, D, 0, st())
    out = torch.empty(2 * D, device=dev)
    L.call("vbx_sum_rows_f32", tmp, Bsz, 2 * D, out, 2 * D, 0, st())
    assert rel_err(out[:D], wd.grad) < 1e-5
    assert rel_err(out[D:], bd.grad) < 1e-5


# ----------------------------------------------------------------------------- text-conditioned embed input
@pytest.mark.parametrize("T", [25, 40, 64])
def test_pack_embed_text_and_table_grad(L, T):
    """[x | to_cond_emb(ids) resized T->N (F.interpolate bilinear, voicebox_pytorch.py:89-107) | cond'] and the scatter of
    d(cond_emb) into the table gradient."""
    Bsz, N, D, E, V = 3, 40, 64, 48, 50
    g = torch.Generator().manual_seed(T)
    x, cond = torch.randn(Bsz, N, D, generator=g), torch.randn(Bsz, N, D, generator=g)
    cmask = torch.rand(Bsz, N, generator=g) < 0.4
    drop = torch.tensor([False, True, False])
    null_cond = torch.randn(D, generator=g)
    ids = torch.randint(0, V, (Bsz, T), generator=g)
    table = torch.randn(V + 1, E, generator=g)
    out16 = torch.empty(Bsz * N, 2 * D + E, dtype=torch.float16, device=dev)
    outb = torch.empty(Bsz * N, 2 * D + E, dtype=torch.bfloat16, device=dev)
    L.call("vbx_pack_embed_input_text", x.to(dev), cond.to(dev), cmask.to(torch.uint8).to(dev), drop.to(torch.uint8).to(dev),
           null_cond.to(dev), ids.to(dev), T, table.to(dev), E, V, out16, outb, Bsz, N, D, st())
    tab = table.double().requires_grad_(True)
    ids_eff = torch.where(drop[:, None], torch.full_like(ids, V), ids)
    emb = tab[ids_eff]
    if T != N:
        emb = F.interpolate(emb.transpose(1, 2)[..., None].float(), (N, 1), mode="bilinear")[..., 0].transpose(1, 2)
        emb_d = F.interpolate(tab[ids_eff].transpose(1, 2)[..., None], (N, 1), mode="bilinear")[..., 0].transpose(1, 2)
    else:
        emb_d = emb
    c2 = cond * (~cmask)[..., None]
    c2 = torch.where(drop[:, None, None], null_cond, c2)
    ref = torch.cat((x, emb.detach().float(), c2), dim=-1).reshape(Bsz * N, -1)
    got = out16.float().cpu()
    assert torch.equal(got[:, :D], x.reshape(-1, D).half().float())
    assert torch.equal(got[:, D + E:], c2.reshape(-1, D).half().float())
    # interpolation weights in fp32 as torch computes them: equal up to the fp16 rounding of the stored operand
    assert max_err(got[:, D:D + E], ref[:, D:D + E]) < 2e-3
    assert rel_err(outb.float(), ref) < 4e-3
    # table gradient
    demb = bf(torch.randn(Bsz * N, E, generator=g))
    gt = torch.zeros(V + 1, E, device=dev)
    L.call("vbx_cond_emb_bwd", demb.to(dev), E, ids.to(dev), T, drop.to(torch.uint8).to(dev), V, gt, Bsz, N, E, st())
    (emb_d.reshape(Bsz * N, E) * demb.double()).sum().backward()
    assert rel_err(gt, tab.grad) < 1e-5, rel_err(gt, tab.grad)


# ----------------------------------------------------------------------------- DurationPredictor front / back end
@pytest.mark.parametrize("S", [17, 28, 40])
def test_pack_phoneme_input_and_rowdot(L, S):
    """[to_phoneme_emb(max(ids,0)) | curtail_or_pad(where(drop, null_cond, cond * ~cond_mask), N)] (voicebox_pytorch.py:793-823),
    bit-exact up to the fp16 rounding of the stored operand; to_pred row dot (:672-675, :833)."""
    Bsz, N, D, E, V = 3, 28, 64, 32, 37
    g = torch.Generator().manual_seed(S)
    cond = torch.randn(Bsz, S, D, generator=g)
    cmask = torch.rand(Bsz, S, generator=g) < 0.4
    null_cond = torch.randn(D, generator=g)
    ids = torch.randint(0, V, (Bsz, N), generator=g)
    ids[1, 20:] = -1
    table = torch.randn(V, E, generator=g)
    for drop in (None, torch.tensor([False, True, False])):
        out16 = torch.empty(Bsz * N, E + D, dtype=torch.float16, device=dev)
        L.call("vbx_pack_phoneme_input", ids.to(dev), table.to(dev), E, cond.to(dev), S, cmask.to(torch.uint8).to(dev),
               drop.to(torch.uint8).to(dev) if drop is not None else None, null_cond.to(dev), out16, Bsz, N, D, st())
        c2 = cond * (~cmask)[..., None]
        if drop is not None:
            c2 = torch.where(drop[:, None, None], null_cond, c2)
        c2 = c2[:, :N] if S > N else F.pad(c2, (0, 0, 0, N - S))
        ref = torch.cat((table[ids.clamp(min=0)], c2), dim=-1).reshape(Bsz * N, -1)
        assert torch.equal(out16.float().cpu(), ref.half().float())
    rows, Dd = 1000, 192
    x, w, b = torch.randn(rows, Dd, generator=g), torch.randn(1, Dd, generator=g), torch.randn(1, generator=g)
    out = torch.empty(rows, device=dev)
    L.call("vbx_rowdot", x.to(dev), w.to(dev), b.to(dev), out, rows, Dd, st())
    assert rel_err(out, x.double() @ w[0].double() + b.double()) < 1e-6
    L.call("vbx_rowdot", x.to(dev), w.to(dev), None, out, rows, Dd, st())
    assert rel_err(out, x.double() @ w[0].double()) < 1e-6


# ----------------------------------------------------------------------------- batched reductions
def test_geglu_bwd_colsum_and_multi_reduce(L):
    """vbx_geglu_bwd_colsum == vbx_geglu_bwd, and its slab records reduced by vbx_multi_reduce (with the GEGLU row
    un-interleave) give the column sums of the stored dh1 = the FeedForward[0].bias gradient."""
    import ctypes as C

    M, Fd, Fp = 333, 170, 192
    g = torch.Generator().manual_seed(12)
    h1 = bf(torch.randn(M, 2 * Fp, generator=g)).to(dev)
    dg = bf(torch.randn(M, Fp, generator=g)).to(dev)
    ref = torch.empty(M, 2 * Fp, dtype=torch.bfloat16, device=dev)
    L.call("vbx_geglu_bwd", h1, dg, ref, M, Fp, st())
    got = torch.empty_like(ref)
    slabs = L.lib().vbx_geglu_bwd_colsum_slabs()
    scratch = torch.zeros(slabs, 2 * Fp, device=dev)
    L.call("vbx_geglu_bwd_colsum", h1, dg, got, M, Fp, scratch, st())
    assert torch.equal(got, ref)

    class Job(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("src_bstride", C.c_long), ("dst_bstride", C.c_long),
                    ("row_stride", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("batches", C.c_int), ("dst_len", C.c_int),
                    ("rowmap", C.c_int), ("F", C.c_int), ("block0", C.c_int), ("pad_", C.c_int)]

    class Jobs(C.Structure):
        _fields_ = [("job", Job * 48), ("n", C.c_int)]  # VBX_MR_MAX

    out = torch.zeros(2 * Fd, device=dev)
    x = torch.randn(3, 37, 100, generator=g).to(dev)  # a second, batched job: per-batch sums over 37 rows of the first 64 columns
    out2 = torch.zeros(3, 80, device=dev)
    jobs = Jobs()
    jobs.n = 2
    j0, j1 = jobs.job[0], jobs.job[1]
    j0.src, j0.dst, j0.rows, j0.cols, j0.row_stride, j0.batches = scratch.data_ptr(), out.data_ptr(), slabs, 2 * Fp, 2 * Fp, 1
    j0.dst_len, j0.rowmap, j0.F = 2 * Fd, 1, Fd
    j1.src, j1.dst, j1.rows, j1.cols, j1.row_stride, j1.batches = x.data_ptr(), out2.data_ptr(), 37, 64, 100, 3
    j1.src_bstride, j1.dst_bstride, j1.dst_len = 37 * 100, 80, 64
    lib = L.lib()
    lib.vbx_multi_reduce.argtypes = [C.POINTER(Jobs), C.c_void_p]
    assert lib.vbx_multi_reduce(C.byref(jobs), st()) == 0, lib.vbx_last_error()
    cs = ref.float().sum(0).view(Fp // 64, 2, 64)
    exp = torch.cat((cs[:, 0].reshape(-1)[:Fd], cs[:, 1].reshape(-1)[:Fd]))
    assert rel_err(out, exp) < 1e-5
    assert rel_err(out2[:, :64], x[:, :, :64].sum(1)) < 1e-6 and float(out2[:, 64:].abs().max()) == 0.0


def test_grouped_splitk_gemm_equals_separate_launches(L):
    """The four weight-gradient shapes of a dim-512 layer (and ragged ones) through ONE grouped launch of the 256 x 256 tile
    (gemm3.hip): the same slabs as four vbx_gemm calls on the 128-wide kernels (same K ranges per split; fp32 accumulation)."""
    g = torch.Generator().manual_seed(0)
    for K, shapes in ((8320, ((3072, 512, 5), (512, 1024, 8), (2816, 512, 5), (512, 1408, 5))),
                      (1000, ((264, 136, 3), (72, 520, 1), (128, 128, 2)))):
        descs = (L.GemmDesc * len(shapes))()
        keep, sep, grp = [], [], []
        for i, (I, J, S) in enumerate(shapes):
            Pm = bf(torch.randn(K, I, generator=g)).to(dev)
            Qm = bf(torch.randn(K, J, generator=g)).to(dev)
            a = torch.full((S, I, J), float("nan"), device=dev)
            b = torch.full((S, I, J), float("nan"), device=dev)
            gemm(L, L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, Pm, Qm, I, J, K, C=a, splits=S)
            d = descs[i]
            d.mode, d.epilogue, d.M, d.N, d.K = L.VBX_GEMM_TN, L.VBX_EPI_SPLITK, I, J, K
            d.A, d.B, d.C, d.lda, d.ldb, d.splits = Pm.data_ptr(), Qm.data_ptr(), b.data_ptr(), I, J, S
            keep += [Pm, Qm]
            sep.append(a)
            grp.append(b)
        rc = L.lib().vbx_gemm_tn_splitk_grouped(descs, len(shapes), st())
        assert rc == 0, L.lib().vbx_last_error()
        torch.cuda.synchronize()
        for a, b in zip(sep, grp):
            assert torch.isfinite(b).all() and rel_err(b, a.double()) < 1e-6


def test_fp16_outputs_saturate(L):
    """forward GEMM operands are stored as fp16 (max 65504).  Outlier activations must clamp, not
    become inf (an inf would give inf * 0 = NaN in the next GEMM): the norm output with a huge gamma, and the v / GEGLU outputs of
    a GEMM with huge weights."""
    D, Bsz, Np = 512, 1, 24
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bsz, Np, D, generator=g)
    gamma = torch.full((D,), 1.0e4)  # normed rows have |x_i| ~ 1 -> y ~ 1e4 .. 4e4 and beyond fp16 range for the tails
    gamma[::7] = 1.0e5
    y16 = torch.empty(Bsz * Np, D, dtype=torch.float16, device=dev)
    L.call("vbx_rmsnorm_fwd", x.to(dev), gamma.to(dev), None, 0, None, y16, Bsz, Np, 0, Np, D, st())
    assert torch.isfinite(y16.float()).all()
    ref = (restate.l2norm_scale(x.double(), D) * gamma.double()).clamp(-65504, 65504).view(Bsz * Np, D)
    assert rel_err(y16, ref) < 1e-3
    assert (y16.float().abs() == 65504).any()  # the clamp was exercised
    # GEGLU output of a GEMM with large weights
    M, K, F = 64, 64, 128
    a = (torch.randn(M, K, generator=g) * 30).half()
    w = (torch.randn(2 * F, K, generator=g) * 30).half()
    bias = torch.zeros(2 * F)
    G = torch.empty(M, F, dtype=torch.float16, device=dev)
    d = L.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = L.VBX_GEMM_NT, L.VBX_EPI_GEGLU, M, 2 * F, K, K, K, F
    ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
    d.A, d.B, d.C, d.bias, d.f16 = ad.data_ptr(), wd.data_ptr(), G.data_ptr(), bd.data_ptr(), 1
    assert L.lib().vbx_gemm(d, st()) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(G.float()).all() and (G.float().abs() == 65504).any()
