"""CPU (-m "not gpu"): host logic of the product package -- no compute calls.
 * the C-ABI library loads and exports every symbol include/vbx.h declares;
 * mask helpers are bit-exact with the reference's golden vectors;
 * state-dict layout equals the reference's (keys, shapes) and parameters flatten into one buffer;
 * the product path FAILS LOUDLY without a GPU (no CPU fallback).
"""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vbx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from voicebox_pytorch_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    l = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) > 40
    missing = [s for s in syms if not hasattr(l, s)]
    assert not missing, missing
    bound = _lib.lib()  # binds prototypes for everything the host calls
    assert bound.vbx_version() == 1
    assert set(_lib.exported_symbols()) <= set(syms) | {"vbx_last_error"}


def test_argument_validation_without_gpu():
    """Entry points validate arguments on the host before any launch: callable without a GPU."""
    from voicebox_pytorch_amd import _lib

    l = _lib.lib()
    d = _lib.GemmDesc()
    assert l.vbx_gemm(d, None) != 0
    assert b"null operand" in l.vbx_last_error()
    with pytest.raises(_lib.VbxError):
        _lib.call("vbx_rmsnorm_fwd", None, None, None, 0, None, None, 1, 1, 0, 1, 64, None)


def test_masks_bit_exact(golden):
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.masks import rng_override

    g = golden("masks")
    for n, expect in g["cases"].items():
        with rng_override(rand=g["rand"]):
            assert torch.equal(vbx.mask_from_frac_lengths(n, g["frac"]), expect), n
    assert torch.equal(vbx.mask_from_start_end_indices(8, g["start"], g["end"]), g["start_end_8"])
    torch.manual_seed(11)  # un-injected path draws from the global generator exactly like the reference (:146)
    assert torch.equal(vbx.mask_from_frac_lengths(1024, g["frac"]), g["frac_helper_1024"])
    assert vbx.prob_mask_like((3,), 1, "cpu").all() and not vbx.prob_mask_like((3,), 0, "cpu").any()
    a, b = torch.tensor([True, False, True]), torch.tensor([True, True, False])
    assert torch.equal(vbx.reduce_masks_with_and(a, None, b), a & b) and vbx.reduce_masks_with_and(None) is None


def test_state_dict_layout_matches_reference(golden):
    import voicebox_pytorch_amd as vbx

    g = golden("small")
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    sd = vb.state_dict()
    assert set(sd) == set(g["state"])
    assert all(sd[k].shape == g["state"][k].shape for k in sd)
    w = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    assert set(w.state_dict()) == {"voicebox." + k for k in sd}
    fp = vb.flat_params()
    assert fp.is_current() and fp.numel >= sum(p.numel() for p in vb.parameters() if p.requires_grad)
    vb.load_state_dict(g["state"])
    assert fp.is_current()
    assert torch.equal(vb.to_pred.weight, g["state"]["to_pred.weight"])
    # gradients complete front-to-back: stage ranges tile the flat buffer contiguously
    lo = 0
    for a, b in fp.stage_ranges:
        assert a == lo and b > a
        lo = b
    assert lo == fp.numel
    # reference parameter count for the benchmark architectures
    n = lambda **kw: sum(p.numel() for p in vbx.VoiceBox(num_cond_tokens=1, dim_head=64, heads=16, condition_on_text=False, **kw).parameters())
    assert n(dim=512, depth=2) == 18654292


def test_no_cpu_fallback(golden):
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd import _lib

    g = golden("small")
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    w = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    if not torch.cuda.is_available():
        with pytest.raises((_lib.VbxError, RuntimeError, AssertionError)):
            w(g["x1"])
        with pytest.raises((_lib.VbxError, RuntimeError, AssertionError)):
            w.sample(cond=g["cond"], steps=3)


def test_unsupported_configurations_raise():
    import voicebox_pytorch_amd as vbx

    for kw in (dict(dim_head=32), dict(conv_pos_embed_kernel_size=33), dict(conv_pos_embed_kernel_size=16), dict(dim=100)):
        base = dict(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
        base.update(kw)
        with pytest.raises((NotImplementedError, AssertionError)):
            vbx.VoiceBox(**base)
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    with pytest.raises(NotImplementedError):
        vbx.ConditionalFlowMatcherWrapper(voicebox=vb, use_torchode=True)


def test_philox_restatement_known_answers_and_dropout_modules():
    """tests/philox_ref.py (the host statement of the dropout mask definition the GPU tests compare the kernels with) reproduces
    the published Philox4x32-10 known-answer vectors (Random123 kat_vectors); a model built with dropout carries the reference's
    parameter-free nn.Dropout holders (attend.py:47, voicebox_pytorch.py:346) and the same state-dict keys as one without."""
    import numpy as np
    import philox_ref as PR
    import voicebox_pytorch_amd as vbx

    kat = (((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)))
    for ctr, key, want in kat:
        got = PR.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    assert PR.thr16(0.1) == 58982 and PR.thr16(0.5) == 32768 and PR.thr16(1e-9) == 65535
    keep = PR.rows_keep(64, 64, 0.25, 12345, 3)
    assert 0.70 < keep.mean() < 0.80
    base = dict(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb = vbx.VoiceBox(**base, attn_dropout=0.1, ff_dropout=0.2)
    assert list(vb.state_dict()) == list(vbx.VoiceBox(**base).state_dict())
    layer = vb.transformer.layers[0]
    assert layer[3].attend.attn_dropout.p == 0.1 and layer[5][2].p == 0.2
    assert vb._cfg["attn_dropout"] == 0.1 and vb._cfg["ff_dropout"] == 0.2
    with pytest.raises(AssertionError):
        vbx.VoiceBox(**base, attn_dropout=1.0)


def test_gateloop_state_dict_layout(golden):
    """use_gateloop_layers=True: the module tree must carry the reference's keys (voicebox_pytorch.py:399) so that
    checkpoints interchange; the flat-buffer order keeps the post-LayerNorm weight|bias contiguous for the runtime."""
    import voicebox_pytorch_amd as vbx

    g = golden("small_gateloop")
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=500, depth=2, dim_head=64, heads=2, condition_on_text=False,
                      use_gateloop_layers=True)
    mine = {k: tuple(v.shape) for k, v in vb.state_dict().items()}
    ref = {k: tuple(v.shape) for k, v in g["state"].items()}
    assert {k: v for k, v in mine.items() if "inv_freq" not in k} == {k: v for k, v in ref.items() if "inv_freq" not in k}
    res = vb.load_state_dict(g["state"], strict=False)
    assert not res.unexpected_keys
    fp = vb.flat_params()
    for l in range(2):
        assert fp.offsets[f"L{l}.GLLNB"] == fp.offsets[f"L{l}.GLLNW"] + 64
    assert torch.equal(vb.transformer.layers[1][1].to_qkva[0].weight, g["state"]["transformer.layers.1.1.to_qkva.0.weight"])


def test_text_conditioned_state_dict(golden):
    import voicebox_pytorch_amd as vbx

    g = golden("small_text")
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=50, dim_cond_emb=48, depth=2, dim_head=64, heads=2, condition_on_text=True)
    mine = {k: tuple(v.shape) for k, v in vb.state_dict().items() if "inv_freq" not in k}
    ref = {k: tuple(v.shape) for k, v in g["state"].items() if "inv_freq" not in k}
    assert mine == ref
    assert vb.null_cond_id == 50 and not vb.null_cond.requires_grad
    w = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    with pytest.raises(NotImplementedError):
        w.sample(cond=g["cond"], phoneme_ids=g["ids"])  # needs a DurationPredictor
    with pytest.raises(Exception):
        w(g["x1"])  # text-conditioned training needs ids (voicebox_pytorch.py:1389); on a CPU-only box the missing GPU raises first


def test_standalone_transformer_state_dict(golden):
    import voicebox_pytorch_amd as vbx

    g = golden("transformer")
    for name, c in g.items():
        tr = vbx.Transformer(dim=64, depth=2, dim_head=64, heads=2, **c["kw"])
        mine = {k: tuple(v.shape) for k, v in tr.state_dict().items() if "inv_freq" not in k}
        ref = {k: tuple(v.shape) for k, v in c["state"].items() if "inv_freq" not in k}
        assert mine == ref, name
        assert not tr.load_state_dict(c["state"], strict=False).unexpected_keys
        if not torch.cuda.is_available():
            with pytest.raises(Exception):  # no CPU fallback
                tr(c["x"], mask=c["mask"], adaptive_rmsnorm_cond=c["cond"])


def test_standalone_transformer_unet_state_dict(golden):
    """use_unet_skip_connection=True: Linear(2 * dim, dim) at layers[i][0] of the second half, None in the first (:394-398)."""
    import voicebox_pytorch_amd as vbx

    g = golden("transformer_unet")
    for name, c in g.items():
        tr = vbx.Transformer(dim=64, depth=4, dim_head=64, heads=2, use_unet_skip_connection=True, **c["kw"])
        mine = {k: tuple(v.shape) for k, v in tr.state_dict().items() if "inv_freq" not in k}
        ref = {k: tuple(v.shape) for k, v in c["state"].items() if "inv_freq" not in k}
        assert mine == ref, name
        assert tr.layers[0][0] is None and tr.layers[1][0] is None and tuple(tr.layers[2][0].weight.shape) == (64, 128)
        assert tr.skip_connect_scale == (c["kw"].get("skip_connect_scale") or 2 ** -0.5)
        assert not tr.load_state_dict(c["state"], strict=False).unexpected_keys
        fp = tr.flat_params()
        assert "L2.SKW" in fp.offsets and "L3.SKB" in fp.offsets and "L1.SKW" not in fp.offsets


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The structs that cross the C ABI by pointer (vbx_model, vbx_io, vbx_gemm_desc, vbx_adam_seg) are declared twice -- include/vbx.h
    and the ctypes mirrors of the host side.  gcc compiles the header and prints sizeof / offsetof of every mirrored field; a
    field added on one side only (or in another place) fails here instead of as silent garbage on the GPU box."""
    import ctypes as C
    import shutil
    import subprocess

    from voicebox_pytorch_amd import _lib, engine

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("vbx_model", engine.VbxModel), ("vbx_io", engine.VbxIO), ("vbx_adam_seg", engine.VbxAdamSeg), ("vbx_gemm_desc", _lib.GemmDesc)]
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "vbx.h"', "int main(void) {"]
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), (cname, got[cname], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_ctypes_prototypes_match_the_header_argument_counts():
    """Every entry point bound in _lib._PROTOS takes as many arguments as include/vbx.h declares for it (ctypes would happily pass too
    few or too many)."""
    import re

    from voicebox_pytorch_amd import _lib

    h = open(os.path.join(ROOT, "include", "vbx.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    h = re.sub(r"//[^\n]*", "", h)
    declared = {}
    for m in re.finditer(r"\b(?:int|size_t|void\*|const char\*|unsigned|long)\s+(vbx_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        declared[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    assert len(declared) > 90
    missing = [k for k in _lib._PROTOS if k not in declared]
    assert not missing, missing
    bad = [(k, declared[k], len(v)) for k, v in _lib._PROTOS.items() if declared[k] != len(v)]
    assert not bad, bad


def test_midpoint_tables_match_oracle_grid():
    """dt_i = t[i+1]-t[i] from the same fp32 linspace as the oracle: linspace(0,1,64) has several distinct dt."""
    t = torch.linspace(0, 1, 64)
    dt = t[1:] - t[:-1]
    assert len(set(dt.tolist())) > 1
    t65 = torch.linspace(0, 1, 65)
    assert len(set((t65[1:] - t65[:-1]).tolist())) == 1  # 64 intervals: dt = 1/64 exactly


def test_interpolate_1d_index_rule():
    """The token->frame resize of the text-conditioned path (csrc/ops.hip:interp_src) restated in numpy float32 and checked
    against F.interpolate(mode='bilinear', align_corners=False) -- what interpolate_1d (voicebox_pytorch.py:89-107) calls."""
    import numpy as np
    import torch.nn.functional as F

    def interp_src(n, N, T):
        if T == N:
            return n, n, np.float32(0)
        scale = np.float32(T) / np.float32(N)
        src = scale * (np.float32(n) + np.float32(0.5)) - np.float32(0.5)
        src = np.float32(0) if src < 0 else src
        i0 = min(int(src), T - 1)
        i1 = i0 + (1 if i0 < T - 1 else 0)
        return i0, i1, np.float32(src - np.float32(i0))

    g = torch.Generator().manual_seed(0)
    for T, N in ((25, 40), (40, 40), (64, 40), (7, 1024), (1000, 96), (1, 5), (5, 1)):
        e = torch.randn(1, 3, T, generator=g)
        ref = F.interpolate(e[..., None], (N, 1), mode="bilinear")[..., 0]
        got = torch.empty(1, 3, N)
        for n in range(N):
            i0, i1, lam = interp_src(n, N, T)
            got[0, :, n] = (1 - float(lam)) * e[0, :, i0] + float(lam) * e[0, :, i1]
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6), (T, N, float((got - ref).abs().max()))


def test_duration_predictor_host_logic(golden):
    """DurationPredictor (voicebox_pytorch.py:596-727) without a GPU: reference state-dict keys load (aligner.* skipped), the
    aligned-phoneme index logic is bit-exact on the reference's durations, unsupported front ends raise, compute refuses the CPU."""
    import voicebox_pytorch_amd as vbx

    g = golden("duration")
    for name, c in g.items():
        dp = vbx.DurationPredictor(num_phoneme_tokens=37, dim_phoneme_emb=32, dim=64, depth=2, dim_head=64, heads=2, **c["kw"])
        sd = dict(c["state"])
        sd["aligner.key_layers.0.weight"] = torch.zeros(2)
        res = dp.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys and all("inv_freq" in k for k in res.missing_keys), res
        mine = {k for k in dp.state_dict() if "inv_freq" not in k}
        assert mine == {k for k in c["state"] if "inv_freq" not in k}
        assert not dp.null_cond.requires_grad
        assert torch.equal(dp.align_phoneme_ids_with_durations(c["ids"], c["d3"]), c["aligned"]), name
        dp.eval()
        with pytest.raises(Exception) as ei:
            dp(cond=c["cond"], phoneme_ids=c["ids"], cond_mask=c["cond_mask"])
        assert "no CPU fallback" in str(ei.value)
        with pytest.raises(NotImplementedError):
            dp(cond=c["cond"], texts=["hello"])
        with pytest.raises(NotImplementedError):
            dp.train()(cond=c["cond"], phoneme_ids=c["ids"])
    with pytest.raises(NotImplementedError):
        vbx.DurationPredictor(dim=64, depth=2, heads=2)  # would need the espeak tokenizer
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=37, depth=2, dim_head=64, heads=2, dim_cond_emb=48, condition_on_text=True)
    cfm = vbx.ConditionalFlowMatcherWrapper(voicebox=vb, duration_predictor=dp)
    assert cfm.duration_predictor is dp
    with pytest.raises(AssertionError):
        vbx.ConditionalFlowMatcherWrapper(voicebox=vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2,
                                                                condition_on_text=False), duration_predictor=dp)


@pytest.mark.parametrize("which", ["gemm3", "gemm4", "epi_stage"])
def test_gemm_tile_lds_layout_emulation(tmp_path, which):
    """csrc/gemm3_layout.hpp / gemm4_layout.hpp (LDS-DMA source permutation, fragment read addresses, transposed-accumulator
    column map of the 256 x 256 and 128 x 256 GEMM tiles; gemm3's K-contiguous layout is also the one-round 64-deep tile's)
    replayed on the host against a plain GEMM, plus bank-conflict freedom of every fragment read and the DS-immediate identities
    the kernels rely on; csrc/epi_stage_layout.hpp (the row-staged epilogues' transposing LDS image): every (row, chunk) comes back
    once and in order, writes at most 2-way, reads conflict free."""
    import subprocess

    exe = str(tmp_path / f"{which}_layout_check")
    src = os.path.join(ROOT, "tests", "native", f"{which}_layout_check.cpp" if which != "epi_stage" else "epi_stage_check.cpp")
    subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:]


def test_weights_key_sees_every_kind_of_parameter_write():
    """ADVICE r1 (high / medium): the packed-operand cache key must change when parameters change through PyTorch (the views have
    their own version counters) and when native code bumps the epoch; a re-flatten must change flat_gen (cached hipGraphs)."""
    import voicebox_pytorch_amd as vbx

    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    fp = vb.flat_params()
    k0, gen0 = fp.weights_key(), fp.flat_gen
    opt = torch.optim.SGD(vb.parameters(), lr=0.1)
    for p in vb.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)
    opt.step()
    k1 = fp.weights_key()
    assert k1 != k0
    vb.load_state_dict(vb.state_dict())
    k2 = fp.weights_key()
    assert k2 != k1
    with torch.no_grad():
        vb.to_pred.weight.mul_(0.5)
    k3 = fp.weights_key()
    assert k3 != k2
    fp.bump()  # what the native Adam / a broadcast into the flat buffer do
    assert fp.weights_key() != k3 and fp.flat_gen == gen0
    # a write through p.data is invisible to every version counter: the documented remedy is mark_weights_dirty()
    k4 = fp.weights_key()
    vb.to_pred.weight.data.mul_(2.0)
    assert fp.weights_key() == k4
    vb.mark_weights_dirty()
    assert fp.weights_key() != k4
    assert vb.flat_params() is fp and fp.is_current()
    vb.double()  # dtype change: the parameters leave the flat buffer -> re-flatten, new generation
    fp2 = vb.flat_params()
    assert fp2.flat_gen == gen0 + 1 and fp2.is_current()


def test_optimizer_state_of_the_other_weight_decay_grouping_is_rejected(tmp_path):
    """a checkpoint written under wd = 0 numbers its optimizer state in one group, a wd > 0 trainer in two
    (optimizer.py:10-35): loading across the two must raise (as torch.optim.load_state_dict does) BEFORE any moment is copied."""
    import types

    from voicebox_pytorch_amd.trainer import VoiceBoxTrainer

    class T:  # just what _load_optim_state_dict touches
        _load_optim_state_dict = VoiceBoxTrainer._load_optim_state_dict

    p = [torch.nn.Parameter(torch.zeros(4, 4)), torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(4, 4))]
    t = T()
    t.wd = 0.1
    t._optim_param_order = lambda: ([p[0], p[2], p[1]], 2)  # decayed (ndim >= 2) first
    fp = types.SimpleNamespace(order=["a", "b", "c"], slots={"a": p[0], "b": p[1], "c": p[2]}, offsets={"a": 0, "b": 16, "c": 20})
    t.train_step_fn = types.SimpleNamespace(fp=fp, m=torch.zeros(36), v=torch.zeros(36), steps=0)
    st = lambda shape: {"step": torch.tensor(3.0), "exp_avg": torch.ones(shape), "exp_avg_sq": torch.ones(shape)}
    one_group = {"state": {0: st((4, 4)), 1: st((4,)), 2: st((4, 4))}, "param_groups": [{"params": [0, 1, 2]}]}
    with pytest.raises(ValueError, match="parameter groups"):
        t._load_optim_state_dict(one_group)
    assert float(t.train_step_fn.m.abs().sum()) == 0.0  # nothing was copied
    wrong_shape = {"state": {0: st((4, 4)), 1: st((4,)), 2: st((4,))}, "param_groups": [{"params": [0, 1]}, {"params": [2]}]}
    with pytest.raises(ValueError, match="shape"):
        t._load_optim_state_dict(wrong_shape)
    assert float(t.train_step_fn.m.abs().sum()) == 0.0
    good = {"state": {0: st((4, 4)), 1: st((4, 4)), 2: st((4,))}, "param_groups": [{"params": [0, 1]}, {"params": [2]}]}
    t._load_optim_state_dict(good)
    assert float(t.train_step_fn.m.sum()) == 36.0 and t.train_step_fn.steps == 3


def test_sampler_split_is_validated(monkeypatch):
    """VBX_SAMPLE_SPLIT / MidpointSampler(split=) accept 1 and 2 only (nothing else is measured or tested)."""
    from voicebox_pytorch_amd import solver

    monkeypatch.setenv("VBX_SAMPLE_SPLIT", "three")
    with pytest.raises(ValueError, match="VBX_SAMPLE_SPLIT"):
        solver.MidpointSampler(None, 8, 16, 3)
    monkeypatch.delenv("VBX_SAMPLE_SPLIT")
    with pytest.raises(ValueError, match="split=4"):
        solver.MidpointSampler(None, 8, 16, 3, split=4)


@pytest.mark.parametrize("depth,served", [(2, True), (12, True), (24, False)])
def test_gradient_norm_range_plan(depth, served):
    """Host side of the clip norm from slab-reduce partials (engine.Engine.sq_partials_info / sumsq_with_adaln_factors): what is read
    again is the complement of the adaLN weight blocks and of every layer's four big weight matrices -- three gaps per layer plus the
    head and the tail.  vbx_sumsq_ranges takes at most 64 ranges of 4-float-aligned bounds: a depth-24 model has 73 gaps and must fall
    back to the plain pass (only the adaLN blocks left out: depth + 1 ranges) instead of failing."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.engine import Engine

    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=depth, dim_head=64, heads=2, condition_on_text=False)
    fp = vb.flat_params()
    n = fp.flat.numel()
    ada, big = [], []
    for l in range(depth):
        ada.append((fp.offsets[f"L{l}.G1W"], fp.offsets[f"L{l}.B2W"] + fp.slots[f"L{l}.B2W"].numel()))
        for k in ("QKVW", "OUTW", "FF1W", "FF2W"):
            big.append((fp.offsets[f"L{l}.{k}"], fp.offsets[f"L{l}.{k}"] + fp.slots[f"L{l}.{k}"].numel()))
    plain = Engine.complement_ranges(n, ada)
    fold = Engine.complement_ranges(n, ada + big)
    assert len(plain) // 2 == depth + 1 <= Engine.SUMSQ_MAX_RANGES
    assert len(fold) // 2 == 3 * depth + 1
    assert (len(fold) // 2 <= Engine.SUMSQ_MAX_RANGES) == served
    for rest, blocks in ((plain, ada), (fold, ada + big)):
        assert all(v % 4 == 0 for v in rest)  # vbx_sumsq_ranges: 16-byte aligned bounds
        covered = sum(hi - lo for lo, hi in blocks) + sum(rest[i + 1] - rest[i] for i in range(0, len(rest), 2))
        assert covered == n and rest == sorted(rest)  # a partition of the buffer
    assert Engine.complement_ranges(10, [(0, 10)]) == [] and Engine.complement_ranges(10, []) == [0, 10]
    assert Engine.complement_ranges(12, [(4, 8), (2, 6)]) == [0, 2, 8, 12]  # overlapping, unsorted blocks


def test_trainer_split_batches_rank_batch():
    """trainer.py:83,93 (Accelerator(split_batches=...)): False -> every rank loads batch_size samples; True -> batch_size is the global
    batch and must be a round multiple of the number of processes."""
    from voicebox_pytorch_amd.trainer import VoiceBoxTrainer

    assert VoiceBoxTrainer.rank_batch(8, 4, False) == 8
    assert VoiceBoxTrainer.rank_batch(8, 4, True) == 2
    assert VoiceBoxTrainer.rank_batch(8, 1, True) == 8
    with pytest.raises(ValueError, match="round multiple"):
        VoiceBoxTrainer.rank_batch(6, 4, True)
