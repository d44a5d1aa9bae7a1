"""DurationPredictor inference on the native kernels.

Mirrors voicebox_pytorch.py:596-839 -- same constructor keywords, module tree and state-dict keys (`to_phoneme_emb`,
`to_embed`, `null_cond`, `conv_embed.dw_conv1d.0`, `transformer.*`, `to_pred.0`) -- for the path the sampler uses
(:1231-1241): `forward` in eval mode, `forward_with_cond_scale` (:694-727) and `align_phoneme_ids_with_durations` (:689-692).
Compute: vbx_pack_phoneme_input (embedding gather + condition masking / dropping / curtail_or_pad, fp16) -> vbx_gemm
(to_embed) -> vbx_convpos_fwd (+ residual) -> the native Transformer stack (plain RMSNorm, no registers) -> vbx_rowdot (to_pred).

Not built, and raising instead of silently differing:
  * training: the reference's training branch (:841-876) needs naturalspeech2_pytorch's `Aligner`, `ForwardSumLoss` and
    `maximum_path` (third-party, absent) and computes its L1 loss on the hidden state rather than the predicted durations;
  * `tokenizer` / `texts` (espeak phonemizer, third-party) and `audio_enc_dec` (codecs are out of scope): pass `phoneme_ids`.
`aligner.*` entries of a reference checkpoint are skipped on load (there is no aligner module here).
"""
from random import random

import torch
from torch import nn

from . import _lib
from .masks import mask_from_frac_lengths, prob_mask_like, take_draw
from .model import ConvPositionEmbed, Transformer, exists


def generate_mask_from_repeats(repeats):
    """naturalspeech2_pytorch's helper (third-party, call site voicebox_pytorch.py:690): repeats (b, i) -> bool (b, i, j) with
    entry set when output position j lies in phoneme i's span [cumsum_i - repeats_i, cumsum_i); j < the longest total."""
    repeats = repeats.int()
    cumsum = repeats.cumsum(dim=-1)
    start = cumsum - repeats
    total = int(cumsum[..., -1].amax().item())
    pos = torch.arange(total, device=repeats.device)
    return (pos >= start[..., None]) & (pos < cumsum[..., None])


class DurationPredictor(nn.Module):
    def __init__(self, *, audio_enc_dec=None, tokenizer=None, num_phoneme_tokens=None, dim_phoneme_emb=512, dim=512, depth=10,
                 dim_head=64, heads=8, ff_mult=4, ff_dropout=0., conv_pos_embed_kernel_size=31, conv_pos_embed_groups=None,
                 attn_dropout=0, attn_flash=False, attn_qk_norm=True, use_gateloop_layers=False, p_drop_prob=0.2,
                 frac_lengths_mask=(0.1, 1.), aligner_kwargs: dict = dict(dim_in=80, attn_channels=80)):
        super().__init__()
        if exists(audio_enc_dec):
            raise NotImplementedError("audio codecs are out of the hot path's scope: feed latents")
        assert not (exists(tokenizer) and exists(num_phoneme_tokens)), \
            'if a phoneme tokenizer was passed into duration module, number of phoneme tokens does not need to be specified'
        if exists(tokenizer) or not exists(num_phoneme_tokens):
            raise NotImplementedError("the espeak phoneme Tokenizer is third-party: pass num_phoneme_tokens and call with phoneme_ids")
        if dim_phoneme_emb % 8 != 0:
            raise NotImplementedError("dim_phoneme_emb must be a multiple of 8 (vectorised embedding gather)")
        # ff_dropout / attn_dropout go to the Transformer as in the reference (:631-642); this module only runs in eval mode (forward
        # raises while .training), where nn.Dropout is the identity, so they never change a result here
        self.audio_enc_dec = None
        self.proj_in = nn.Identity()
        self.tokenizer = None
        self.to_phoneme_emb = nn.Embedding(num_phoneme_tokens, dim_phoneme_emb)
        self.p_drop_prob = p_drop_prob
        self.frac_lengths_mask = frac_lengths_mask
        self.to_embed = nn.Linear(dim + dim_phoneme_emb, dim)
        self.null_cond = nn.Parameter(torch.zeros(dim), requires_grad=False)
        self.conv_embed = ConvPositionEmbed(dim=dim, kernel_size=conv_pos_embed_kernel_size, groups=conv_pos_embed_groups)
        self.transformer = Transformer(dim=dim, depth=depth, dim_head=dim_head, heads=heads, ff_mult=ff_mult, ff_dropout=ff_dropout,
                                       attn_dropout=attn_dropout, attn_flash=attn_flash, attn_qk_norm=attn_qk_norm,
                                       use_gateloop_layers=use_gateloop_layers)
        self.to_pred = nn.Sequential(nn.Linear(dim, 1), nn.Identity())  # [1]: Rearrange('... 1 -> ...'), done by vbx_rowdot
        self.dim, self.dim_phoneme_emb, self.ksize = dim, dim_phoneme_emb, conv_pos_embed_kernel_size
        self.aligner = None  # naturalspeech2_pytorch.Aligner(dim_hidden=dim_phoneme_emb, **aligner_kwargs): training only
        self.align_loss = None

    @property
    def device(self):
        return next(self.parameters()).device

    def load_state_dict(self, state_dict, strict=True, **kw):
        kept = {k: v for k, v in state_dict.items() if not k.startswith("aligner.")}
        return super().load_state_dict(kept, strict=strict, **kw)

    def align_phoneme_ids_with_durations(self, phoneme_ids, durations):  # voicebox_pytorch.py:689-692
        repeat_mask = generate_mask_from_repeats(durations.clamp(min=1))
        # einsum('b i, b i j -> b j') of the reference: every output position belongs to at most one phoneme
        return torch.einsum('bi,bij->bj', phoneme_ids.float(), repeat_mask.float()).long()

    @torch.inference_mode()
    def forward_with_cond_scale(self, *args, texts=None, phoneme_ids=None, cond_scale=1., return_aligned_phoneme_ids=False,
                                **kwargs):  # voicebox_pytorch.py:694-727
        if exists(texts):
            raise NotImplementedError("texts need the espeak Tokenizer (third-party): pass phoneme_ids")
        fk = dict(return_aligned_phoneme_ids=False, phoneme_ids=phoneme_ids)
        durations = self.forward(*args, cond_drop_prob=0., **fk, **kwargs)
        if cond_scale != 1.:
            null_durations = self.forward(*args, cond_drop_prob=1., **fk, **kwargs)
            durations = null_durations + (durations - null_durations) * cond_scale
        if not return_aligned_phoneme_ids:
            return durations
        return durations, self.align_phoneme_ids_with_durations(phoneme_ids.to(durations.device), durations)

    def forward(self, *, cond, texts=None, phoneme_ids=None, cond_drop_prob=0., target=None, cond_mask=None, mel=None,
                phoneme_len=None, mel_len=None, phoneme_mask=None, mel_mask=None, self_attn_mask=None,
                return_aligned_phoneme_ids=False):  # voicebox_pytorch.py:757-839
        if self.training:
            raise NotImplementedError(
                "DurationPredictor training needs naturalspeech2_pytorch's Aligner / maximum_path (third-party, absent); "
                "call .eval() -- the inference path is what ConditionalFlowMatcherWrapper.sample uses")
        if exists(texts) or not exists(phoneme_ids):
            raise NotImplementedError("texts need the espeak Tokenizer (third-party): pass phoneme_ids")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.VbxError("DurationPredictor compute runs only on an MI355X (gfx950) through libvbx_hip.so; "
                                f"parameters are on '{dev}' and there is no CPU fallback")
        with torch.no_grad():
            cond = cond.to(dev, torch.float32).contiguous()
            batch, seq_len, cond_dim = cond.shape
            assert cond_dim == self.dim
            ids = phoneme_ids.to(dev, torch.long).contiguous()
            assert ids.ndim == 2 and ids.shape[0] == batch
            n = ids.shape[-1]
            if not exists(cond_mask):  # :786-791
                coin = take_draw("coin")
                if (random() < 0.5) if coin is None else bool(coin):
                    frac = take_draw("frac_lengths")
                    if frac is None:
                        frac = torch.zeros((batch,), device=dev).float().uniform_(*self.frac_lengths_mask)
                    cond_mask = mask_from_frac_lengths(seq_len, frac.to(dev))
                else:
                    cond_mask = prob_mask_like((batch, seq_len), self.p_drop_prob, dev)
            cmask = cond_mask.to(dev).to(torch.uint8).contiguous()
            drop = None
            if cond_drop_prob > 0.:  # :797-804
                drop = take_draw("cond_drop")
                drop = prob_mask_like((batch,), cond_drop_prob, dev) if drop is None else drop.to(dev)
                drop = drop.to(torch.uint8).contiguous()
            if not exists(self_attn_mask):
                self_attn_mask = ids != -1  # :808-809 (phoneme id -1 is padding)
            amask = self_attn_mask.to(dev).to(torch.bool)
            am8 = amask.to(torch.uint8).contiguous()
            E, D = self.dim_phoneme_emb, self.dim
            st = _lib.current_stream()
            packed = torch.empty(batch * n, E + D, dtype=torch.float16, device=dev)
            _lib.call("vbx_pack_phoneme_input", ids, self.to_phoneme_emb.weight.detach().float().contiguous(), E, cond, seq_len,
                      cmask, drop, self.null_cond.detach().float().contiguous(), packed, batch, n, D, st)
            w16 = self.to_embed.weight.detach().to(torch.float16).contiguous()
            bias = self.to_embed.bias.detach().float().contiguous()
            e = torch.empty(batch * n, D, dtype=torch.float32, device=dev)
            d = _lib.GemmDesc()
            d.mode, d.epilogue, d.M, d.N, d.K = _lib.VBX_GEMM_NT, _lib.VBX_EPI_F32, batch * n, D, E + D
            d.lda, d.ldb, d.ldc, d.f16 = E + D, E + D, D, 1
            d.A, d.B, d.C, d.bias = packed.data_ptr(), w16.data_ptr(), e.data_ptr(), bias.data_ptr()
            rc = _lib.lib().vbx_gemm(d, st)  # to_embed (:823-824)
            if rc != 0:
                raise _lib.VbxError(f"vbx_gemm failed (rc={rc}): {_lib.lib().vbx_last_error().decode()}")
            conv = self.conv_embed.dw_conv1d[0]
            x = torch.empty(batch, n, D, dtype=torch.float32, device=dev)
            _lib.call("vbx_convpos_fwd", e, conv.weight.detach().float().contiguous(), conv.bias.detach().float().contiguous(), am8,
                      None, x, batch, n, 0, D, self.ksize, st)  # conv_embed(x, mask) + x (:826)
            hid = self.transformer(x, mask=amask).contiguous()  # :828-831
            pred = self.to_pred[0]
            durations = torch.empty(batch, n, dtype=torch.float32, device=dev)
            _lib.call("vbx_rowdot", hid, pred.weight.detach().float().contiguous(), pred.bias.detach().float().contiguous(), durations,
                      batch * n, D, st)  # :833
        if not return_aligned_phoneme_ids:
            return durations
        return durations, self.align_phoneme_ids_with_durations(ids.clamp(min=0), durations)  # ids clamped as :811 does before :839
