"""Drop-in model classes: same constructors, call signatures and state-dict layout as the reference
(voicebox_pytorch.py:878-1427, attend.py:38-137), with the compute done by libvbx_hip.so.

There is NO eager/PyTorch fallback for the compute: forward on a non-gfx950 device or without the
shared library raises.  The sub-modules below (Linear, Conv1d, ...) only *hold* parameters so that
`state_dict()` keys/shapes equal the reference's and checkpoints interchange.
"""
import math
from pathlib import Path

import weakref

import torch
from torch import nn

from . import _lib
from .engine import Engine, FlatParams, precise_enabled
from .masks import mask_from_frac_lengths, prob_mask_like, reduce_masks_with_and, take_draw


def exists(v):
    return v is not None


def default(v, d):
    return v if exists(v) else d


# --------------------------------------------------------------------------------------- Attend
def attn_dropout_bits(B, H, Np, p, seed, stream_id, device):
    """Keep bits of one attention call in the two orientations the kernels read (vbx_attn_dropout_bits): int32 tensors
    [B*H, Np, W] row-major (bit key % 32 of word key // 32) and column-major (bit q % 32 of word q // 32)."""
    W = _lib.lib().vbx_dropout_bits_words(Np)
    rm = torch.empty(B * H, Np, W, dtype=torch.int32, device=device)
    cm = torch.empty_like(rm)
    _lib.call("vbx_attn_dropout_bits", rm, cm, B * H, Np, int(seed), int(stream_id), float(p), _lib.current_stream())
    return rm, cm


class _AttendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, drop_p=0., drop_seed=0):
        B, H, Np, dh = q.shape
        dev = q.device
        # kernel contract (include/vbx.h): q16 carries scale * log2(e), so q16 . k16 is the exponent of exp2; qb stays unscaled
        q16 = (q.float() * _lib.lib().vbx_attn_q_prescale(float(scale))).to(torch.float16).contiguous()
        k16 = k.to(torch.float16).contiguous()
        qb = q.to(torch.float16).to(torch.bfloat16).contiguous()
        v16 = v.to(torch.float16).contiguous()
        vb = v.to(torch.bfloat16).contiguous()
        m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
        out16 = torch.empty(B, Np, H * 64, dtype=torch.float16, device=dev)
        lse = torch.empty(B, H, Np, dtype=torch.float32, device=dev)
        none = torch.empty(0, device=dev)
        rm = cm = none
        if drop_p > 0.:  # attend.py:131 -- the keep bits are saved for the backward (1 bit per score, both orientations)
            rm, cm = attn_dropout_bits(B, H, Np, drop_p, drop_seed, 0, dev)
            _lib.call("vbx_attn_fwd_dropout", q16, k16, v16, m8, out16, None, lse, B, H, Np, float(scale), rm, float(drop_p),
                      _lib.current_stream())
        else:
            _lib.call("vbx_attn_fwd", q16, k16, v16, m8, out16, None, lse, B, H, Np, float(scale), _lib.current_stream())
        ctx.save_for_backward(q16, k16, vb, out16, lse, m8 if m8 is not None else none, rm, cm, qb)
        ctx.has_mask, ctx.scale, ctx.in_dtype, ctx.drop_p = m8 is not None, float(scale), q.dtype, float(drop_p)
        return out16.view(B, Np, H, 64).permute(0, 2, 1, 3).to(q.dtype)

    @staticmethod
    def backward(ctx, dout):
        q16, k16, vb, out, lse, m8, rm, cm, qb = ctx.saved_tensors
        B, H, Np, _ = q16.shape
        dev = q16.device
        do = dout.permute(0, 2, 1, 3).reshape(B, Np, H * 64).to(torch.bfloat16).contiguous()
        kb = k16.to(torch.bfloat16)
        delta = torch.empty(B, H, Np, dtype=torch.float32, device=dev)
        dq = torch.empty(B, H, Np, 64, dtype=torch.float32, device=dev)
        dk = torch.empty_like(dq)
        dv = torch.empty(B, Np, H * 64, dtype=torch.bfloat16, device=dev)
        if ctx.drop_p > 0.:
            _lib.call("vbx_attn_bwd_dropout", q16, k16, qb, kb, vb, m8 if ctx.has_mask else None, out, 1, do, lse, delta, dq, dk, dv,
                      H * 64, B, H, Np, ctx.scale, rm, cm, ctx.drop_p, _lib.current_stream())
        else:
            _lib.call("vbx_attn_bwd", q16, k16, qb, kb, vb, m8 if ctx.has_mask else None, out, 1, do, lse, delta, dq, dk, dv,
                      H * 64, B, H, Np, ctx.scale, None, _lib.current_stream())  # (`scratch`: unused since round 6)
        dvh = dv.view(B, Np, H, 64).permute(0, 2, 1, 3)
        return dq.to(ctx.in_dtype), dk.to(ctx.in_dtype), dvh.to(ctx.in_dtype), None, None, None, None


class Attend(nn.Module):
    """attend.py:38-137.  `flash` is accepted for API compatibility: both reference paths compute the same
    function and this implementation is always the fused HIP kernel."""

    def __init__(self, dropout=0., flash=False, scale=None):
        super().__init__()
        assert 0. <= dropout < 1.
        self.dropout, self.flash, self.scale = dropout, flash, scale
        self.attn_dropout = nn.Dropout(dropout)  # attend.py:49: a parameter-free holder; the mask is drawn inside the kernels
        self.last_dropout_seed = None

    def forward(self, q, k, v, mask=None):
        if q.shape[-1] != 64:
            raise NotImplementedError("the HIP attention kernel is built for dim_head == 64")
        if exists(mask) and mask.ndim == 4:
            # attend.py:113-114 rearranges a (b, j) key-padding mask to (b, 1, 1, j) and lets a 4-D mask through unchanged; the kernels
            # take key masks, so a 4-D mask must BE one: size 1 along heads and queries (batch 1 broadcasts)
            if mask.shape[1] != 1 or mask.shape[2] != 1 or mask.shape[0] not in (1, q.shape[0]) or mask.shape[3] != k.shape[2]:
                raise NotImplementedError("4-D attention masks are supported in their key-padding form (b | 1, 1, 1, keys) only; "
                                          f"got {tuple(mask.shape)} (per-head / per-query masks are not built)")
            mask = mask[:, 0, 0, :].expand(q.shape[0], -1)
        if exists(mask) and mask.ndim != 2:
            raise NotImplementedError("only (batch, keys) and (batch, 1, 1, keys) key-padding masks are supported")
        scale = default(self.scale, q.shape[-1] ** -0.5)
        if self.training and self.dropout > 0.:  # one Philox key per call from torch's CPU generator (torch.manual_seed reproduces it)
            self.last_dropout_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            return _AttendFn.apply(q, k, v, mask, scale, float(self.dropout), self.last_dropout_seed)
        return _AttendFn.apply(q, k, v, mask, scale)


# --------------------------------------------------------------------------------------- parameter holders
class LearnedSinusoidalPosEmb(nn.Module):  # voicebox_pytorch.py:154-167
    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))


class RotaryEmbedding(nn.Module):  # voicebox_pytorch.py:172-191
    def __init__(self, dim, theta=50000):
        super().__init__()
        self.theta = theta
        self.register_buffer("inv_freq", 1.0 / (theta ** (torch.arange(0, dim, 2).float() / dim)))


class ConvPositionEmbed(nn.Module):  # voicebox_pytorch.py:203-233
    def __init__(self, dim, *, kernel_size, groups=None):
        super().__init__()
        assert kernel_size % 2 == 1
        groups = default(groups, dim)
        if groups != dim:
            raise NotImplementedError("only the full depthwise conv positional embedding (groups == dim) is implemented")
        self.dw_conv1d = nn.Sequential(nn.Conv1d(dim, dim, kernel_size, groups=groups, padding=kernel_size // 2), nn.GELU())


class RMSNorm(nn.Module):  # voicebox_pytorch.py:237-247
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim))


class AdaptiveRMSNorm(nn.Module):  # voicebox_pytorch.py:249-276
    def __init__(self, dim, cond_dim=None):
        super().__init__()
        cond_dim = default(cond_dim, dim)
        self.scale = dim ** 0.5
        self.to_gamma = nn.Linear(cond_dim, dim)
        self.to_beta = nn.Linear(cond_dim, dim)
        nn.init.zeros_(self.to_gamma.weight)
        nn.init.ones_(self.to_gamma.bias)
        nn.init.zeros_(self.to_beta.weight)
        nn.init.zeros_(self.to_beta.bias)


class MultiheadRMSNorm(nn.Module):  # voicebox_pytorch.py:280-287
    def __init__(self, dim, heads):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(heads, 1, dim))


class Attention(nn.Module):  # voicebox_pytorch.py:289-333
    def __init__(self, dim, dim_head=64, heads=8, dropout=0, flash=False, qk_norm=False, qk_norm_scale=10):
        super().__init__()
        self.heads = heads
        dim_inner = dim_head * heads
        self.attend = Attend(dropout, flash=flash, scale=qk_norm_scale if qk_norm else None)
        self.qk_norm = qk_norm
        if qk_norm:
            self.q_norm = MultiheadRMSNorm(dim_head, heads=heads)
            self.k_norm = MultiheadRMSNorm(dim_head, heads=heads)
        self.to_qkv = nn.Linear(dim, dim_inner * 3, bias=False)
        self.to_out = nn.Linear(dim_inner, dim, bias=False)


class GEGLU(nn.Module):  # voicebox_pytorch.py:337-340 (fused into the FF-in GEMM epilogue)
    pass


def FeedForward(dim, mult=4, dropout=0.):  # voicebox_pytorch.py:342-349 (parameter holder; the dropout runs in the stage runtime)
    assert 0. <= dropout < 1.
    dim_inner = int(dim * mult * 2 / 3)
    return nn.Sequential(nn.Linear(dim, dim_inner * 2), GEGLU(), nn.Dropout(dropout), nn.Linear(dim_inner, dim))


class GateLoop(nn.Module):
    """Parameter holder of gateloop_transformer.SimpleGateLoopLayer(dim, post_ln=True) (third-party; call sites
    voicebox_pytorch.py:31,399,465-466): state-dict keys norm.gamma, to_qkva.0.weight, maybe_post_ln.{weight,bias}.
    The compute (RMSNorm -> Linear(D, 3D) -> gated linear scan -> LayerNorm) is csrc/gateloop.hip."""

    def __init__(self, dim, use_jax_associative_scan=False, post_ln=True):
        super().__init__()
        if use_jax_associative_scan:
            raise NotImplementedError("gateloop_use_jax: there is no jax in the HIP path (the scan is a native kernel)")
        assert post_ln
        self.norm = RMSNorm(dim)
        self.to_qkva = nn.Sequential(nn.Linear(dim, dim * 3, bias=False))
        self.maybe_post_ln = nn.LayerNorm(dim)


class _StackFn(torch.autograd.Function):
    """Transformer.forward as ONE autograd node over the native stack entry points (vbx_model.stack_only)."""

    @staticmethod
    def forward(ctx, tr, eng, x, cond, mask, *params):
        out = eng.forward_stack(x, cond, mask)
        ctx.tr, ctx.eng, ctx.gen, ctx.has_cond = tr, eng, eng.generation, cond is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        tr, eng = ctx.tr, ctx.eng
        if eng.generation != ctx.gen:
            raise RuntimeError("Transformer: another forward of the same (batch, frames) shape ran before this backward; the "
                               "activation arena holds one training forward at a time")
        gflat = torch.zeros(tr._flat.numel, dtype=torch.float32, device=eng.device)
        dx, dcond = eng.backward_stack(gflat, dout)
        return (None, None, dx, dcond if ctx.has_cond else None, None) + tuple(tr._flat.grad_views(gflat))


class Transformer(nn.Module):
    """voicebox_pytorch.py:353-479, same module tree / state-dict keys.  Inside VoiceBox the stack runs fused in
    vbx_model_forward; called on its own (`Transformer.forward(x, mask, adaptive_rmsnorm_cond)`, :412-479) it runs the same
    native layer sequence through the stack-only mode of the runtime (registers + x in, final RMSNorm out)."""

    def __init__(self, dim, *, depth, dim_head=64, heads=8, ff_mult=4, attn_dropout=0., ff_dropout=0.,
                 num_register_tokens=0., attn_flash=False, adaptive_rmsnorm=False, adaptive_rmsnorm_cond_dim_in=None,
                 use_unet_skip_connection=False, skip_connect_scale=None, attn_qk_norm=False, use_gateloop_layers=False,
                 gateloop_use_jax=False):
        super().__init__()
        assert depth % 2 == 0
        self.layers = nn.ModuleList([])
        self.rotary_emb = RotaryEmbedding(dim=dim_head)
        self.num_register_tokens = int(num_register_tokens)
        self.has_register_tokens = num_register_tokens > 0
        if self.has_register_tokens:
            self.register_tokens = nn.Parameter(torch.randn(int(num_register_tokens), dim))
        self.adaptive_rmsnorm = adaptive_rmsnorm
        if dim_head != 64:
            raise NotImplementedError("the HIP kernels are built for dim_head == 64")
        if dim % 64 != 0 or dim > 2048 or heads % 2 != 0:
            raise NotImplementedError("dim must be a multiple of 64 (<= 2048) and heads even")
        cond_dim = default(adaptive_rmsnorm_cond_dim_in, dim)  # AdaptiveRMSNorm: cond_dim = default(cond_dim, dim) (:256)
        self._cfg = dict(D=dim, H=heads, L=depth, F=int(dim * ff_mult * 2 / 3), Th=cond_dim if adaptive_rmsnorm else 8,
                         R=int(num_register_tokens), ksize=31, qk_norm=bool(attn_qk_norm),
                         attn_scale=10.0 if attn_qk_norm else dim_head ** -0.5, theta=50000.0,
                         gateloop=bool(use_gateloop_layers), stack_only=True, plain_norm=not adaptive_rmsnorm,
                         attn_dropout=float(attn_dropout), ff_dropout=float(ff_dropout),
                         unet=bool(use_unet_skip_connection), skip_scale=float(default(skip_connect_scale, 2 ** -0.5)))
        self._flat = None
        self._engines = {}
        norm = (lambda: AdaptiveRMSNorm(dim, cond_dim=adaptive_rmsnorm_cond_dim_in)) if adaptive_rmsnorm else (lambda: RMSNorm(dim))
        self.skip_connect_scale = default(skip_connect_scale, 2 ** -0.5)
        for ind in range(depth):
            has_skip = use_unet_skip_connection and (ind + 1) > (depth // 2)  # :394-398: the second half combines with the first half's inputs
            self.layers.append(nn.ModuleList([
                nn.Linear(dim * 2, dim) if has_skip else None,
                GateLoop(dim=dim, use_jax_associative_scan=gateloop_use_jax) if use_gateloop_layers else None, norm(),
                Attention(dim=dim, dim_head=dim_head, heads=heads, dropout=attn_dropout, flash=attn_flash, qk_norm=attn_qk_norm),
                norm(), FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout)]))
        self.final_norm = RMSNorm(dim)

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- native plumbing of the standalone call (VoiceBox owns its own flat buffer over the same parameters)
    def _layer_slots(self, s):
        for l, layer in enumerate(self.layers):
            skip, gl, n1, attn, n2, ff = layer
            p = f"L{l}."
            if skip is not None:
                s[p + "SKW"], s[p + "SKB"] = skip.weight, skip.bias
            if gl is not None:
                s[p + "GLG"], s[p + "GLW"] = gl.norm.gamma, gl.to_qkva[0].weight
                s[p + "GLLNW"], s[p + "GLLNB"] = gl.maybe_post_ln.weight, gl.maybe_post_ln.bias
            if self.adaptive_rmsnorm:
                s[p + "G1W"], s[p + "G1B"] = n1.to_gamma.weight, n1.to_gamma.bias
                s[p + "B1W"], s[p + "B1B"] = n1.to_beta.weight, n1.to_beta.bias
                s[p + "G2W"], s[p + "G2B"] = n2.to_gamma.weight, n2.to_gamma.bias
                s[p + "B2W"], s[p + "B2B"] = n2.to_beta.weight, n2.to_beta.bias
            else:
                s[p + "N1G"], s[p + "N2G"] = n1.gamma, n2.gamma
            if attn.qk_norm:
                s[p + "QG"], s[p + "KG"] = attn.q_norm.gamma, attn.k_norm.gamma
            s[p + "QKVW"], s[p + "OUTW"] = attn.to_qkv.weight, attn.to_out.weight
            s[p + "FF1W"], s[p + "FF1B"], s[p + "FF2W"], s[p + "FF2B"] = ff[0].weight, ff[0].bias, ff[3].weight, ff[3].bias
        return s

    def _slots(self):
        s = {"FNG": self.final_norm.gamma}
        if self.has_register_tokens:
            s["REG"] = self.register_tokens
        return self._layer_slots(s)

    def flat_params(self):
        if self._flat is None:
            self._flat = FlatParams(self._slots(), self._cfg["L"])
        if not self._flat.is_current():
            self._flat.slots = self._slots()
            self._flat.flatten()
            self._engines.clear()
        return self._flat

    def forward(self, x, mask=None, adaptive_rmsnorm_cond=None):  # voicebox_pytorch.py:412-479
        if precise_enabled():  # the process-wide precise switch must not silently return fast-path results here
            raise NotImplementedError("precise mode (set_precise / precise_mode / VBX_PRECISE) covers VoiceBox only; the stand-alone "
                                      "Transformer (and DurationPredictor) run on the fast path -- leave precise mode to call them")
        if self.adaptive_rmsnorm:
            assert exists(adaptive_rmsnorm_cond), "adaptive_rmsnorm = True needs adaptive_rmsnorm_cond (batch, cond_dim)"
        else:
            assert not exists(adaptive_rmsnorm_cond), "this Transformer was built with adaptive_rmsnorm = False"
        fp = self.flat_params()
        dev = fp.flat.device
        if dev.type != "cuda":
            raise _lib.VbxError("Transformer compute runs only on an MI355X (gfx950) through libvbx_hip.so; "
                                f"parameters are on '{dev}' and there is no CPU fallback")
        B, N, D = x.shape
        assert D == self._cfg["D"]
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                                or (exists(adaptive_rmsnorm_cond) and adaptive_rmsnorm_cond.requires_grad))
        key = (B, N, bool(training))
        eng = self._engines.get(key)
        if eng is None:
            if len(self._engines) >= 4:
                self._engines.pop(next(iter(self._engines)))
            eng = Engine(self._cfg, fp, B, N, training, dev)
            self._engines[key] = eng
        eng.dropout_active = self.training  # nn.Dropout: the module's mode, not the autograd mode
        x32 = x.to(dev, torch.float32)
        c32 = adaptive_rmsnorm_cond.to(dev, torch.float32) if exists(adaptive_rmsnorm_cond) else None
        if exists(c32):
            assert c32.shape == (B, self._cfg["Th"]), (tuple(c32.shape), (B, self._cfg["Th"]))
        m = mask.to(dev) if exists(mask) else None
        if not training:
            return eng.forward_stack(x32, c32, m)
        return _StackFn.apply(self, eng, x32, c32, m, *[fp.slots[s] for s in fp.order])


# --------------------------------------------------------------------------------------- VoiceBox
class _InFlight:
    """Token of a training forward whose backward has not run yet (held by its autograd node, weakly referenced by the engine)."""
    __slots__ = ("__weakref__",)


def _arena_busy(eng):
    ref = getattr(eng, "_in_flight", None)
    return ref is not None and ref() is not None


class _VoiceBoxLossFn(torch.autograd.Function):
    """loss = VoiceBox(w, target=flow) as ONE autograd node: forward = vbx_model_forward, backward =
    vbx_model_backward_{head,layer,embed} into a fresh flat gradient buffer whose views are returned per parameter."""

    @staticmethod
    def forward(ctx, vb, eng, x, cond, cond_mask, times, attn_mask, target, loss_mask, text, *params):
        loss = eng.forward(x, cond, cond_mask, times, attn_mask=attn_mask, target=target, loss_mask=loss_mask, text=text)
        ctx.vb, ctx.eng, ctx.gen = vb, eng, eng.generation
        ctx.wkey = (eng.wpack_owner or eng).packed_version  # the operand copies this forward ran on (shared by the slot engines)
        ctx.token = _InFlight()           # dies with the autograd graph: the arena is free again when nobody can call backward
        eng._in_flight = weakref.ref(ctx.token)
        return loss.clone().reshape(())

    @staticmethod
    def backward(ctx, gloss):
        vb, eng = ctx.vb, ctx.eng
        if eng.generation != ctx.gen:
            raise RuntimeError("VoiceBox: two later forwards of the same (batch, frames) shape ran before this backward; the two "
                               "activation arenas of a shape hold the two most recent training forwards")
        if (eng.wpack_owner or eng).packed_version != ctx.wkey:
            # the reference's autograd raises "modified by an inplace operation" here; with two arenas sharing one set of packed
            # weights a later forward would otherwise silently hand this backward the NEW bf16 weights beside the OLD activations
            raise RuntimeError("VoiceBox: parameters were modified (and re-packed by a later forward) between this forward and its "
                               "backward; run backward before updating the weights")
        gflat = torch.zeros(vb._flat.numel, dtype=torch.float32, device=eng.device)
        gscale = gloss.detach().to(torch.float32).reshape(1).contiguous()
        eng.backward(gflat, gscale=gscale)
        eng._in_flight = None
        return (None,) * 10 + tuple(vb._flat.grad_views(gflat))


class VoiceBox(nn.Module):
    def __init__(self, *, num_cond_tokens=None, audio_enc_dec=None, dim_in=None, dim_cond_emb=1024, dim=1024, depth=24,
                 dim_head=64, heads=16, ff_mult=4, ff_dropout=0., time_hidden_dim=None, conv_pos_embed_kernel_size=31,
                 conv_pos_embed_groups=None, attn_dropout=0, attn_flash=False, attn_qk_norm=True, use_gateloop_layers=False,
                 num_register_tokens=16, p_drop_prob=0.3, frac_lengths_mask=(0.7, 1.), condition_on_text=True):
        super().__init__()
        dim_in = default(dim_in, dim)
        time_hidden_dim = default(time_hidden_dim, dim * 4)
        assert not (condition_on_text and not exists(num_cond_tokens)), \
            'number of conditioning tokens must be specified (whether phonemes or semantic token ids) if training conditional voicebox'
        if condition_on_text and dim_cond_emb % 8 != 0:
            raise NotImplementedError("dim_cond_emb must be a multiple of 8 (16-byte GEMM rows)")
        if exists(audio_enc_dec):
            raise NotImplementedError("audio codecs are out of scope of the hot path: feed latents directly")
        if dim_in % 8 != 0:  # data width of x / cond / target / pred (e.g. 80 mel bins into a dim-512 model, :905,938,964)
            raise NotImplementedError("dim_in must be a multiple of 8 (16-byte rows of the to_embed / to_pred operands)")
        if dim_head != 64:
            raise NotImplementedError("the HIP kernels are built for dim_head == 64")
        if dim % 64 != 0 or dim > 2048 or heads % 2 != 0:
            raise NotImplementedError("dim must be a multiple of 64 (<= 2048) and heads even")
        if conv_pos_embed_kernel_size % 2 != 1 or not 1 <= conv_pos_embed_kernel_size <= 31:
            raise NotImplementedError("conv_pos_embed_kernel_size must be odd (as in the reference, :211) and <= 31")
        self.audio_enc_dec = None
        self.proj_in = nn.Identity()
        self.sinu_pos_emb = nn.Sequential(LearnedSinusoidalPosEmb(dim), nn.Linear(dim, time_hidden_dim), nn.SiLU())
        if not condition_on_text:  # voicebox_pytorch.py:922-926
            dim_cond_emb = 0
        self.dim_cond_emb = dim_cond_emb
        self.condition_on_text = condition_on_text
        self.num_cond_tokens = num_cond_tokens
        if condition_on_text:  # :931-934
            self.null_cond_id = num_cond_tokens  # last token id is the null token of classifier-free guidance
            self.to_cond_emb = nn.Embedding(num_cond_tokens + 1, dim_cond_emb)
        self.p_drop_prob = p_drop_prob
        self.frac_lengths_mask = frac_lengths_mask
        self.to_embed = nn.Linear(dim_in * 2 + dim_cond_emb, dim)
        self.null_cond = nn.Parameter(torch.zeros(dim_in), requires_grad=False)
        self.conv_embed = ConvPositionEmbed(dim=dim, kernel_size=conv_pos_embed_kernel_size, groups=conv_pos_embed_groups)
        self.transformer = Transformer(dim=dim, depth=depth, dim_head=dim_head, heads=heads, ff_mult=ff_mult,
                                       ff_dropout=ff_dropout, attn_dropout=attn_dropout, attn_flash=attn_flash,
                                       attn_qk_norm=attn_qk_norm, num_register_tokens=num_register_tokens,
                                       adaptive_rmsnorm=True, adaptive_rmsnorm_cond_dim_in=time_hidden_dim,
                                       use_gateloop_layers=use_gateloop_layers)
        self.to_pred = nn.Linear(dim, dim_in, bias=False)
        self._cfg = dict(D=dim, H=heads, L=depth, F=int(dim * ff_mult * 2 / 3), Th=time_hidden_dim,
                         R=int(num_register_tokens), ksize=conv_pos_embed_kernel_size, qk_norm=bool(attn_qk_norm),
                         attn_scale=10.0 if attn_qk_norm else dim_head ** -0.5, theta=50000.0,
                         gateloop=bool(use_gateloop_layers), E=dim_cond_emb,
                         V1=(num_cond_tokens + 1) if condition_on_text else 0,
                         attn_dropout=float(attn_dropout), ff_dropout=float(ff_dropout), Din=int(dim_in))
        self._flat = None
        self._engines = {}

    # ---- native plumbing
    def _slots(self):
        t = self.transformer
        s = {"SINW": self.sinu_pos_emb[0].weights, "T1W": self.sinu_pos_emb[1].weight, "T1B": self.sinu_pos_emb[1].bias,
             "EMBW": self.to_embed.weight, "EMBB": self.to_embed.bias, "CONVW": self.conv_embed.dw_conv1d[0].weight,
             "CONVB": self.conv_embed.dw_conv1d[0].bias, "FNG": t.final_norm.gamma, "PREDW": self.to_pred.weight}
        if t.has_register_tokens:
            s["REG"] = t.register_tokens
        if self.condition_on_text:
            s["CEMB"] = self.to_cond_emb.weight
        t._layer_slots(s)
        return s

    def flat_params(self):
        """Ensures every trainable parameter is a view into one flat fp32 buffer on the model's device."""
        if self._flat is None:
            self._flat = FlatParams(self._slots(), self._cfg["L"])
        if not self._flat.is_current():
            self._flat.slots = self._slots()
            self._flat.flatten()
            self._engines.clear()
        return self._flat

    def engine(self, B, N, training, slot=0, wpack_from=None):
        """slot > 0: a further engine (own activation arena) for the same shape -- the sampler integrates the halves of a
        batch concurrently on two streams."""
        fp = self.flat_params()
        dev = fp.flat.device
        if dev.type != "cuda":
            raise _lib.VbxError("VoiceBox compute runs only on an MI355X (gfx950) through libvbx_hip.so; "
                                f"parameters are on '{dev}' and there is no CPU fallback")
        precise = precise_enabled()
        # key[2] carries the mode: 0/1 = inference/training on the fast path, 2/3 = the same in precise mode (own arenas)
        tkey = int(bool(training)) + (2 if precise else 0)
        key = (B, N, tkey) if slot == 0 else (B, N, tkey, slot)
        eng = self._engines.get(key)
        if eng is not None and eng.wpack_owner is not wpack_from:
            eng = None
        if eng is None:
            # Arenas are large (3.6 GB for dim 512 / depth 12 / 8 x 1024 in training), so only FOUR shapes are kept.  A shape = an
            # owner engine (key of length 3) plus the slot engines that share its packed weights (the sampler's second half batch:
            # one more activation arena); they are evicted TOGETHER -- a slot engine must not outlive its owner in the cache, and an
            # owner is not dropped while a sibling slot still shares its arena of packed weights.
            owners = [k for k in self._engines if len(k) == 3]
            if len(owners) >= 4 and key[:3] not in owners:
                victim = next(k for k in owners if self._engines[k] is not wpack_from)
                for k in [k for k in self._engines if k[:3] == victim]:
                    self._engines.pop(k)
            eng = Engine(self._cfg, fp, B, N, training, dev, wpack_from=wpack_from, precise=precise)
            self._engines[key] = eng
        eng.dropout_active = self.training  # nn.Dropout semantics (attend.py:131, voicebox_pytorch.py:346): the module's mode
        return eng

    def mark_weights_dirty(self):
        """Call after writing parameter values through `p.data` (p.data.copy_(), EMA weight swaps, ...): such writes bump neither the
        parameter's nor the flat buffer's version counter, so nothing else tells the engines -- and the hipGraph samplers holding
        them -- that their fp16 / bf16 packed operand copies are stale.  Writes through the parameters themselves (optimizer steps,
        load_state_dict, p.copy_()) and this package's own training step are detected automatically."""
        self.flat_params().bump()

    @property
    def device(self):
        return next(self.parameters()).device

    @torch.inference_mode()
    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):  # voicebox_pytorch.py:972-985
        logits = self.forward(*args, cond_drop_prob=0., **kwargs)
        if cond_scale == 1.:
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1., **kwargs)
        return null_logits + (logits - null_logits) * cond_scale

    def forward(self, x, *, times, cond_token_ids, self_attn_mask=None, cond_drop_prob=0.1, target=None, cond=None,
                cond_mask=None):  # voicebox_pytorch.py:987-1115
        cond = default(cond, target)  # :1003
        assert exists(cond), "either cond or target must be given"
        batch, seq_len, cond_dim = cond.shape
        assert cond_dim == x.shape[-1]
        dev = self.device
        x, cond = x.to(dev, torch.float32), cond.to(dev, torch.float32)
        times = torch.as_tensor(times, device=dev).to(torch.float32)
        if times.ndim == 0:  # :1015-1019 (odeint hands a 0-dim time)
            times = times.expand(batch)
        if times.ndim == 1 and times.shape[0] == 1:
            times = times.expand(batch)
        times = times.contiguous()
        if self.training:  # :1023-1029
            if not exists(cond_mask):
                frac = take_draw("frac_lengths")
                if frac is None:
                    frac = torch.zeros((batch,), device=dev).float().uniform_(*self.frac_lengths_mask)
                cond_mask = mask_from_frac_lengths(seq_len, frac.to(dev))
        elif not exists(cond_mask):
            cond_mask = torch.ones((batch, seq_len), device=dev, dtype=torch.bool)
        cond_mask = cond_mask.to(dev)
        # classifier-free guidance drop (:1040-1053) and the conditioning token ids (:1055-1066)
        drop = None
        if cond_drop_prob > 0.:
            if not self.condition_on_text:
                # same failure as the reference on an unconditional model (:1050-1053)
                raise AttributeError("'VoiceBox' object has no attribute 'null_cond_id'")
            drop = take_draw("cond_drop")
            drop = prob_mask_like((batch,), cond_drop_prob, dev) if drop is None else drop.to(dev)
        if exists(self_attn_mask):
            self_attn_mask = self_attn_mask.to(dev)
        text = None
        if self.condition_on_text:
            assert exists(cond_token_ids), "a text-conditioned VoiceBox needs cond_token_ids (batch, tokens)"
            ids = cond_token_ids.to(dev)
            if ids.shape[-1] != seq_len and exists(self_attn_mask) and self_attn_mask.shape[-1] != seq_len:
                # :1064-1066 (interpolate_1d on the boolean mask, the reference's own torch ops)
                m4 = self_attn_mask.float()[:, None, :, None]
                self_attn_mask = torch.nn.functional.interpolate(m4, (seq_len, 1), mode="bilinear")[:, 0, :, 0].to(torch.bool)
            text = (ids, self.null_cond_id, drop, self.null_cond)
        if not exists(target):
            eng = self.engine(batch, seq_len, training=False)
            return eng.forward(x, cond, cond_mask, times, attn_mask=self_attn_mask, text=text)
        target = target.to(dev, torch.float32)
        loss_mask = reduce_masks_with_and(cond_mask, self_attn_mask)  # :1099
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # The reference's autograd keeps any number of graphs alive (voicebox_pytorch.py:1416-1425); an activation arena holds ONE
            # training forward.  Two arenas per shape form a ring: a second forward before the first one's backward (two micro-batch
            # losses summed before .backward(), a regulariser evaluated on a second input) takes the other arena, which shares the
            # first one's packed weights.  A third forward while both are still waiting takes over the OLDER arena: that graph's
            # backward then raises (as every second forward did before round 3) -- forwards themselves never fail.
            eng = self.engine(batch, seq_len, training=True)
            if _arena_busy(eng):
                eng1 = self.engine(batch, seq_len, training=True, slot=1, wpack_from=eng)
                if not _arena_busy(eng1) or getattr(eng1, "_flight_seq", 0) < getattr(eng, "_flight_seq", 0):
                    eng = eng1
            self._flight_counter = getattr(self, "_flight_counter", 0) + 1
            eng._flight_seq = self._flight_counter
            fp = self._flat
            params = [fp.slots[s] for s in fp.order]
            return _VoiceBoxLossFn.apply(self, eng, x, cond, cond_mask, times, self_attn_mask, target, loss_mask, text, *params)
        eng = self.engine(batch, seq_len, training=False)
        return eng.forward(x, cond, cond_mask, times, attn_mask=self_attn_mask, target=target, loss_mask=loss_mask,
                           text=text).clone().reshape(())


# --------------------------------------------------------------------------------------- CFM wrapper
def is_probably_audio_from_shape(t):  # voicebox_pytorch.py:1119-1120
    return exists(t) and (t.ndim == 2 or (t.ndim == 3 and t.shape[1] == 1))


class ConditionalFlowMatcherWrapper(nn.Module):
    def __init__(self, voicebox, text_to_semantic=None, duration_predictor=None, sigma=0., ode_atol=1e-5, ode_rtol=1e-5,
                 use_torchode=False, torchdiffeq_ode_method='midpoint', torchode_method_klass=None, cond_drop_prob=0.):
        super().__init__()
        assert isinstance(voicebox, VoiceBox)
        self.sigma = sigma
        self.voicebox = voicebox
        self.condition_on_text = voicebox.condition_on_text
        assert not (not self.condition_on_text and exists(text_to_semantic)), \
            'TextToSemantic should not be passed in if not conditioning on text'
        if exists(text_to_semantic):
            raise NotImplementedError("TextToSemantic (spear-tts) is a third-party front end, out of scope: pass semantic_token_ids")
        if exists(duration_predictor):
            from .duration import DurationPredictor

            assert isinstance(duration_predictor, DurationPredictor)
            assert self.condition_on_text, 'a duration predictor aligns phoneme ids for a text-conditioned VoiceBox'
        if use_torchode:
            raise NotImplementedError("the adaptive torchode/Tsit5 path is replaced by the built-in fixed-step midpoint solver")
        if torchdiffeq_ode_method != 'midpoint':
            raise NotImplementedError("only the fixed-grid midpoint method (the reference default) is built")
        self.text_to_semantic = None
        self.duration_predictor = duration_predictor  # a submodule, as in the reference (:1147): its weights are in state_dict()
        self.cond_drop_prob = cond_drop_prob
        self.use_torchode = False
        self.odeint_kwargs = dict(atol=ode_atol, rtol=ode_rtol, method=torchdiffeq_ode_method)  # atol/rtol: unused by fixed grids
        self._samplers = {}

    @property
    def device(self):
        return next(self.parameters()).device

    def load(self, path, strict=True):  # voicebox_pytorch.py:1167-1173
        path = Path(path)
        assert path.exists()
        pkg = torch.load(str(path), map_location='cpu')
        self.load_state_dict(pkg['model'], strict=strict)
        return pkg

    @torch.inference_mode()
    def sample(self, *, cond=None, texts=None, text_token_ids=None, semantic_token_ids=None, phoneme_ids=None,
               cond_mask=None, steps=3, cond_scale=1., decode_to_audio=True, decode_to_codes=False,
               max_semantic_token_ids=2048, spec_decode=False, spec_decode_gamma=5, use_graph=True):
        """voicebox_pytorch.py:1175-1330 with torchdiffeq's fixed-grid midpoint replaced by the built-in solver
        (solver.py): `steps` time points on linspace(0,1,steps) -> 2*(steps-1) function evaluations."""
        from .solver import MidpointSampler

        if is_probably_audio_from_shape(cond):
            raise NotImplementedError("raw-audio conditioning needs an audio codec (out of scope)")
        num_cond_inputs = sum(map(exists, (texts, text_token_ids, semantic_token_ids, phoneme_ids)))
        assert num_cond_inputs <= 1
        cond_token_ids = None
        dev = self.device
        if self.condition_on_text:  # :1208-1255
            if exists(texts) or exists(text_token_ids):
                raise NotImplementedError("text -> semantic tokens needs a TextToSemantic module (out of scope): pass semantic_token_ids")
            if exists(semantic_token_ids):
                assert not exists(phoneme_ids)
                cond_token_ids = semantic_token_ids.to(dev)
            else:  # :1231-1241: phoneme ids -> predicted durations -> one id per frame
                assert exists(phoneme_ids), "a text-conditioned model samples from semantic_token_ids or phoneme_ids"
                if not exists(self.duration_predictor):
                    raise NotImplementedError("phoneme_ids need a DurationPredictor (pass duration_predictor=) to be aligned to frames")
                assert exists(cond), "the duration predictor is conditioned on cond (B, frames, dim)"
                self.duration_predictor.eval()
                _, cond_token_ids = self.duration_predictor.forward_with_cond_scale(
                    cond=cond, phoneme_ids=phoneme_ids, return_aligned_phoneme_ids=True)
            target_len = cond_token_ids.shape[-1]
            if exists(cond):  # curtail_or_pad(cond, cond_target_length) (:109-119, :1253)
                n = cond.shape[-2]
                cond = cond[..., :target_len, :] if n > target_len else torch.nn.functional.pad(cond, (0, 0, 0, target_len - n))
            else:
                raise NotImplementedError("cond = None needs audio_enc_dec.latent_dim (codecs are out of scope): pass cond")
        else:
            assert num_cond_inputs == 0, 'no conditioning inputs should be given if not conditioning on text'
            if cond_scale != 1.:
                # reference: second pass with cond_drop_prob = 1 -> AttributeError on an unconditional model
                raise AttributeError("'VoiceBox' object has no attribute 'null_cond_id'")
        assert exists(cond), "cond (B, frames, dim) is required"
        self.voicebox.eval()
        cond = cond.to(dev, torch.float32)
        y0 = take_draw("y0")
        y0 = torch.randn_like(cond) if y0 is None else y0.to(dev, torch.float32)
        B, N, _ = cond.shape
        T = cond_token_ids.shape[-1] if exists(cond_token_ids) else 0
        key = (B, N, steps, bool(use_graph), T, float(cond_scale) != 1., precise_enabled())
        fp = self.voicebox.flat_params()
        # a re-flatten (.to(), dtype change) frees the buffer whose addresses the cached hipGraphs captured: drop them
        self._samplers = {k: s for k, s in self._samplers.items() if s.flat_gen == fp.flat_gen and s.eng.fp is fp}
        smp = self._samplers.get(key)
        if smp is None:
            if len(self._samplers) >= 2:
                self._samplers.pop(next(iter(self._samplers)))
            smp = MidpointSampler(self.voicebox, B, N, steps, use_graph=use_graph, tokens=T, guided=float(cond_scale) != 1.)
            self._samplers[key] = smp
        return smp.run(y0, cond, cond_mask, cond_token_ids=cond_token_ids, cond_scale=float(cond_scale))

    def forward(self, x1, *, mask=None, semantic_token_ids=None, phoneme_ids=None, cond=None, cond_mask=None,
                input_sampling_rate=None):  # voicebox_pytorch.py:1332-1427
        if is_probably_audio_from_shape(x1) or is_probably_audio_from_shape(cond):
            assert exists(self.voicebox.audio_enc_dec), 'audio_enc_dec must be set on VoiceBox to train directly on raw audio'
        assert self.condition_on_text or not (exists(semantic_token_ids) or exists(phoneme_ids)), \
            'semantic or phoneme ids should not be passed in if not conditioning on text'
        dev = self.device
        x1 = x1.to(dev, torch.float32).contiguous()
        batch = x1.shape[0]
        x0 = take_draw("x0")
        x0 = torch.randn_like(x1) if x0 is None else x0.to(dev, torch.float32)  # :1399
        times = take_draw("times")
        times = torch.rand((batch,), dtype=x1.dtype, device=dev) if times is None else times.to(dev, torch.float32)  # :1403
        w = torch.empty_like(x1)
        flow = torch.empty_like(x1)
        _lib.call("vbx_cfm_inputs", x1, x0.contiguous(), times.contiguous(), float(self.sigma), w, flow, batch,
                  x1[0].numel(), _lib.current_stream())  # :1408-1410
        cond_token_ids = None
        if self.condition_on_text:  # :1374-1390 (no TextToSemantic module here: ids are given)
            if exists(semantic_token_ids):
                assert not exists(phoneme_ids), 'phoneme ids are not needed for conditioning with spear-tts text-to-semantic'
                cond_token_ids = semantic_token_ids
            else:
                assert exists(phoneme_ids)
                cond_token_ids = phoneme_ids
        self.voicebox.train()  # :1414
        return self.voicebox(w, cond=cond, cond_mask=cond_mask, times=times, target=flow, self_attn_mask=mask,
                             cond_token_ids=cond_token_ids, cond_drop_prob=self.cond_drop_prob)
