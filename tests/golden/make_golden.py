"""Generates the committed golden vectors by running the UNMODIFIED reference
(/root/reference, imported through oracle/ref_loader.py with third-party stubs).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
Outputs (committed): tests/golden/{masks,rotary,small,small_gateloop,small_text,transformer,duration,cfg1,cfg4,small_wc,cfg4_wc,cfg5_wc,cfg4_wc_train,cfg4_seeds,cfg3,cfg5_wc_b8,small_dropout,small_dimin,cfg4_seeds_exact,attend_mask4d,transformer_unet,init_stats}.pt

RNG protocol: the reference draws, from the global CPU generator,
randn_like(x1) -> rand(B) -> uniform_(0.7,1)(B) -> uniform_(0,1)(B) per training
step; we replay the same calls under the same seed to record the injected values.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_loader, restate  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def replay_draws(x1, seed):
    torch.manual_seed(seed)
    b = x1.shape[0]
    x0 = torch.randn_like(x1)
    times = torch.rand((b,), dtype=x1.dtype)
    frac = torch.zeros((b,)).float().uniform_(0.7, 1.0)
    rand = torch.zeros_like(frac).float().uniform_(0, 1)
    return x0, times, frac, rand


def build_reference(ref, cfg, state=None, seed=0):
    torch.manual_seed(seed)
    vb = ref.VoiceBox(dim=cfg.dim, num_cond_tokens=500, depth=cfg.depth, dim_head=cfg.dim_head,
                      heads=cfg.heads, condition_on_text=False,
                      num_register_tokens=cfg.num_register_tokens, use_gateloop_layers=cfg.use_gateloop)
    wrapper = ref.ConditionalFlowMatcherWrapper(voicebox=vb)
    if state is not None:
        missing = vb.load_state_dict(state, strict=False)
        # only the non-persistent-free buffer inv_freq may be absent from our recipe dict
        assert all(k.endswith("rotary_emb.inv_freq") for k in missing.missing_keys), missing
        assert not missing.unexpected_keys, missing
    return vb, wrapper


def gen_masks(ref):
    fr = torch.tensor([0.7, 0.85, 0.9999, 1.0, 0.7123, 0.93])
    rd = torch.tensor([0.0, 0.5, 0.999, 0.3, 0.77, 1.0 - 1e-7])
    out = {"frac": fr, "rand": rd, "cases": {}}
    for n in (8, 37, 1024):
        lengths = (fr * n).long()
        start = ((n - lengths) * rd).clamp(min=0)
        out["cases"][n] = ref.mask_from_start_end_indices(n, start, start + lengths)
    s = torch.tensor([2.9, 0.0, 7.2])
    e = torch.tensor([5.9, 0.99, 8.0])
    out["start"], out["end"] = s, e
    out["start_end_8"] = ref.mask_from_start_end_indices(8, s, e)
    # the full helper with its own RNG draw, replayed
    torch.manual_seed(11)
    out["frac_helper_1024"] = ref.mask_from_frac_lengths(1024, fr)
    torch.manual_seed(11)
    out["frac_helper_rand"] = torch.zeros_like(fr).float().uniform_(0, 1)
    torch.save(out, os.path.join(HERE, "masks.pt"))


def gen_rotary(ref):
    rot = ref.RotaryEmbedding(dim=64)
    pos = torch.cat((torch.full((16,), -10000, dtype=torch.long), torch.arange(48)))
    freqs = rot(pos)
    t = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    torch.save({"positions": pos, "freqs": freqs, "t": t, "rotated": ref.apply_rotary_pos_emb(freqs, t)},
               os.path.join(HERE, "rotary.pt"))


def gen_small(ref):
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    vb, wrapper = build_reference(ref, cfg, seed=0)
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():  # randomise the zero-initialised adaLN projections
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("_norm.gamma") or name.endswith("final_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    b, n = 2, 40
    x1 = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(7))
    x0, times, frac, rand = replay_draws(x1, seed=99)
    torch.manual_seed(99)
    loss = wrapper(x1)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    # padded-batch variant: key-padding / loss mask given
    mask = torch.ones(b, n, dtype=torch.bool)
    mask[1, 29:] = False
    vb.zero_grad()
    torch.manual_seed(99)
    loss_m = wrapper(x1, mask=mask)
    loss_m.backward()
    grads_m = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    # eval forward (prediction) with an explicit cond / cond_mask
    vb.eval()
    cond = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(8))
    cmask = torch.zeros(b, n, dtype=torch.bool)
    cmask[:, 10:30] = True
    tt = torch.tensor([0.25, 0.8])
    with torch.no_grad():
        pred = vb(x1, times=tt, cond_token_ids=None, cond=cond, cond_mask=cmask, cond_drop_prob=0.0)
        pred_scalar_t = vb(x1, times=torch.tensor(0.5), cond_token_ids=None, cond=cond, cond_drop_prob=0.0)
    # sampling, steps=3 and steps=5 with the :1289 draw replayed
    torch.manual_seed(3)
    y0 = torch.randn_like(cond)
    torch.manual_seed(3)
    s3 = wrapper.sample(cond=cond, steps=3)
    torch.manual_seed(3)
    s5 = wrapper.sample(cond=cond, steps=5)
    torch.save(dict(cfg=dict(dim=64, depth=2, heads=2, dim_head=64), state=state, x1=x1, x0=x0, times=times,
                    frac=frac, rand=rand, loss=loss.detach(), grads=grads, mask=mask, loss_masked=loss_m.detach(),
                    grads_masked=grads_m, cond=cond, cond_mask=cmask, eval_times=tt, pred=pred,
                    pred_scalar_t=pred_scalar_t, y0=y0, sample3=s3, sample5=s5),
               os.path.join(HERE, "small.pt"))
    print("small: loss", float(loss), "masked", float(loss_m))


def gen_small_gateloop(ref):
    """use_gateloop_layers=True: the reference's module tree and call order (voicebox_pytorch.py:399,465-466) around
    the RESTATED third-party layer (oracle/restate.py:GateLoopRestated -- gateloop_transformer is not installable
    here, so the layer's own arithmetic is parity-unpinned; its placement, residual and state-dict keys are pinned)."""
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64, use_gateloop=True)
    vb, wrapper = build_reference(ref, cfg, seed=0)
    g = torch.Generator().manual_seed(321)
    with torch.no_grad():
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("norm.gamma") or ".maybe_post_ln." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    b, n = 2, 72
    x1 = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(17))
    x0, times, frac, rand = replay_draws(x1, seed=98)
    torch.manual_seed(98)
    loss = wrapper(x1)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor([0.3, 0.6]), cond_token_ids=None, cond=x1, cond_drop_prob=0.0)
    torch.save(dict(cfg=dict(dim=64, depth=2, heads=2, dim_head=64, use_gateloop=True), state=state, x1=x1, x0=x0,
                    times=times, frac=frac, rand=rand, loss=loss.detach(), grads=grads, eval_times=torch.tensor([0.3, 0.6]),
                    pred=pred), os.path.join(HERE, "small_gateloop.pt"))
    print("small_gateloop: loss", float(loss))


def gen_small_text(ref):
    """condition_on_text=True (voicebox_pytorch.py:931-940, 1039-1076, 972-985): phoneme ids with tokens != frames (bilinear
    resize), classifier-free-guidance drop during training, guided sampling from semantic ids."""
    torch.manual_seed(0)
    vb = ref.VoiceBox(dim=64, num_cond_tokens=50, dim_cond_emb=48, depth=2, dim_head=64, heads=2, condition_on_text=True,
                      num_register_tokens=16)
    wrapper = ref.ConditionalFlowMatcherWrapper(voicebox=vb, cond_drop_prob=0.5)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("_norm.gamma") or name.endswith("final_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
        vb.null_cond.add_(torch.randn(vb.null_cond.shape, generator=g) * 0.3)  # exercise the null_cond substitution
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    b, n, t = 3, 40, 25
    x1 = torch.randn(b, n, 64, generator=g)
    ids = torch.randint(0, 50, (b, t), generator=g)
    x0, times, frac, rand = replay_draws(x1, seed=55)
    drop = torch.zeros((b,)).float().uniform_(0, 1) < 0.5  # prob_mask_like (:68-74), the draw after the span-mask draws
    torch.manual_seed(55)
    loss = wrapper(x1, phoneme_ids=ids)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    # tokens == frames, no drop
    wrapper.cond_drop_prob = 0.
    ids_n = torch.randint(0, 50, (b, n), generator=g)
    vb.zero_grad()
    torch.manual_seed(55)
    loss_n = wrapper(x1, semantic_token_ids=ids_n)
    loss_n.backward()
    grads_n = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    cond = torch.randn(b, n, 64, generator=g)
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor(0.4), cond_token_ids=ids, cond=cond, cond_drop_prob=0.0)
        pred_cfg = vb.forward_with_cond_scale(x1, times=torch.tensor(0.4), cond_token_ids=ids, cond=cond, cond_scale=1.7)
    torch.manual_seed(4)
    y0 = torch.randn_like(cond)
    torch.manual_seed(4)
    s3 = wrapper.sample(cond=cond, semantic_token_ids=ids_n, steps=3, cond_scale=1.3)
    torch.save(dict(state=state, x1=x1, ids=ids, ids_n=ids_n, x0=x0, times=times, frac=frac, rand=rand, drop=drop, loss=loss.detach(),
                    grads=grads, loss_n=loss_n.detach(), grads_n=grads_n, cond=cond, pred=pred, pred_cfg=pred_cfg, y0=y0, sample3=s3),
               os.path.join(HERE, "small_text.pt"))
    print("small_text: loss", float(loss), "drop", drop.tolist(), "loss_n", float(loss_n))


def gen_transformer(ref):
    """Standalone Transformer.forward (voicebox_pytorch.py:412-479): adaptive (registers, qk-norm, key-padding mask) and plain
    (no registers, no qk-norm) variants, output + gradients of parameters, input and condition."""
    out = {}
    for name, kw, use_mask in (("adaptive", dict(num_register_tokens=4, adaptive_rmsnorm=True, adaptive_rmsnorm_cond_dim_in=32,
                                                attn_qk_norm=True), True),
                               ("plain", dict(num_register_tokens=0, adaptive_rmsnorm=False, attn_qk_norm=False), False)):
        torch.manual_seed(5)
        tr = ref.Transformer(dim=64, depth=2, dim_head=64, heads=2, **kw)
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            for n, prm in tr.named_parameters():
                if ".to_gamma." in n or ".to_beta." in n:
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
                if n.endswith("gamma"):
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
        state = {k: v.detach().clone() for k, v in tr.state_dict().items()}
        b, n = 2, 40
        x = torch.randn(b, n, 64, generator=g).requires_grad_(True)
        cond = torch.randn(b, 32, generator=g).requires_grad_(True) if kw["adaptive_rmsnorm"] else None
        mask = None
        if use_mask:
            mask = torch.ones(b, n, dtype=torch.bool)
            mask[1, 31:] = False
        dout = torch.randn(b, n, 64, generator=g)
        y = tr(x, mask=mask, adaptive_rmsnorm_cond=cond)
        (y * dout).sum().backward()
        out[name] = dict(kw=kw, state=state, x=x.detach().clone(), cond=None if cond is None else cond.detach().clone(), mask=mask,
                         dout=dout, y=y.detach().clone(), dx=x.grad.clone(), dcond=None if cond is None else cond.grad.clone(),
                         grads={k: p.grad.detach().clone() for k, p in tr.named_parameters() if p.grad is not None})
        print("transformer", name, float(y.norm()))
    torch.save(out, os.path.join(HERE, "transformer.pt"))


def gen_transformer_unet(ref):
    """Standalone Transformer with use_unet_skip_connection=True (voicebox_pytorch.py:368-369,391-398,453-463), depth 4 = two skips:
    adaptive (registers, qk-norm, key-padding mask, skip_connect_scale=0.5), plain (default scale 2^-0.5) and plain + GateLoop
    layers (the restated third-party layer, see gen_small_gateloop): output + gradients of parameters, input and condition."""
    out = {}
    for name, kw, use_mask in (("adaptive", dict(num_register_tokens=4, adaptive_rmsnorm=True, adaptive_rmsnorm_cond_dim_in=32,
                                                attn_qk_norm=True, skip_connect_scale=0.5), True),
                               ("plain", dict(num_register_tokens=0, adaptive_rmsnorm=False, attn_qk_norm=False), False),
                               ("gateloop", dict(num_register_tokens=2, adaptive_rmsnorm=False, attn_qk_norm=True,
                                                 use_gateloop_layers=True), False)):
        torch.manual_seed(6)
        tr = ref.Transformer(dim=64, depth=4, dim_head=64, heads=2, use_unet_skip_connection=True, **kw)
        g = torch.Generator().manual_seed(12)
        with torch.no_grad():
            for n, prm in tr.named_parameters():
                if ".to_gamma." in n or ".to_beta." in n:
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
                if n.endswith("gamma"):
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
                if "q_norm.gamma" in n or "k_norm.gamma" in n:
                    prm.mul_(0.25)  # trained-regime logits (std ~5): at depth 4 the random-init softmax is chaotic
        state = {k: v.detach().clone() for k, v in tr.state_dict().items()}
        assert "layers.2.0.weight" in state and "layers.3.0.bias" in state and "layers.1.0.weight" not in state
        b, n = 2, 40
        x = torch.randn(b, n, 64, generator=g).requires_grad_(True)
        cond = torch.randn(b, 32, generator=g).requires_grad_(True) if kw["adaptive_rmsnorm"] else None
        mask = None
        if use_mask:
            mask = torch.ones(b, n, dtype=torch.bool)
            mask[1, 31:] = False
        dout = torch.randn(b, n, 64, generator=g)
        y = tr(x, mask=mask, adaptive_rmsnorm_cond=cond)
        (y * dout).sum().backward()
        out[name] = dict(kw=kw, state=state, x=x.detach().clone(), cond=None if cond is None else cond.detach().clone(), mask=mask,
                         dout=dout, y=y.detach().clone(), dx=x.grad.clone(), dcond=None if cond is None else cond.grad.clone(),
                         grads={k: p.grad.detach().to(torch.bfloat16) for k, p in tr.named_parameters() if p.grad is not None})  # direction check only
        print("transformer_unet", name, float(y.norm()))
    torch.save(out, os.path.join(HERE, "transformer_unet.pt"))


def gen_duration(ref):
    """DurationPredictor in eval mode (voicebox_pytorch.py:596-839; the Aligner / ForwardSumLoss members are third-party
    placeholders, never called at inference): durations with and without classifier-free guidance, ragged phoneme padding
    (-1), cond shorter AND longer than the phoneme sequence (curtail_or_pad), and the aligned phoneme ids (:689-692, which goes
    through the restated third-party generate_mask_from_repeats -- parity unpinned for that helper)."""
    out = {}
    for name, S, kw in (("short_cond", 21, dict(attn_qk_norm=True)), ("long_cond", 40, dict(attn_qk_norm=False))):
        torch.manual_seed(9)
        dp = ref.DurationPredictor(num_phoneme_tokens=37, dim_phoneme_emb=32, dim=64, depth=2, dim_head=64, heads=2, **kw)
        g = torch.Generator().manual_seed(13)
        with torch.no_grad():
            for n, prm in dp.named_parameters():
                if n.endswith("gamma"):
                    prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
            dp.null_cond.copy_(torch.randn(64, generator=g) * 0.5)  # a checkpoint may carry any value; exercise the drop path
            dp.to_pred[0].weight.mul_(4.0)
            dp.to_pred[0].bias.add_(2.5)  # durations around 2.5 frames so that clamp / int() / repeats are exercised
        dp.eval()
        state = {k: v.detach().clone() for k, v in dp.state_dict().items()}
        b, n = 3, 28
        ids = torch.randint(0, 37, (b, n), generator=g)
        ids[1, 20:] = -1
        ids[2, 9:] = -1
        cond = torch.randn(b, S, 64, generator=g)
        cond_mask = torch.rand(b, S, generator=g) < 0.3
        with torch.no_grad():
            d1 = dp(cond=cond, phoneme_ids=ids, cond_mask=cond_mask)
            d_null = dp(cond=cond, phoneme_ids=ids, cond_mask=cond_mask, cond_drop_prob=1.0)
            d3, aligned = dp.forward_with_cond_scale(cond=cond, phoneme_ids=ids, cond_mask=cond_mask, cond_scale=3.0,
                                                     return_aligned_phoneme_ids=True)
        out[name] = dict(kw=kw, state=state, ids=ids, cond=cond, cond_mask=cond_mask, d1=d1.clone(), d_null=d_null.clone(),
                         d3=d3.clone(), aligned=aligned.clone())
        print("duration", name, d1[0, :6].tolist(), tuple(aligned.shape))
    torch.save(out, os.path.join(HERE, "duration.pt"))


def gen_cfg1(ref):
    """BASELINE config 1/2: dim 512, depth 2, heads 16, B=2, N=1024.  Weights by the committed
    recipe oracle.restate.init_state_dict(seed=0) (too big to commit); only scalars/slices stored."""
    cfg = restate.Cfg(dim=512, depth=2, heads=16, dim_head=64)
    state = restate.init_state_dict(cfg, seed=0)
    vb, wrapper = build_reference(ref, cfg, state=state)
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(0))
    x0, times, frac, rand = replay_draws(x1, seed=1)
    torch.manual_seed(1)
    loss = wrapper(x1)
    loss.backward()
    gnorm = {k: float(p.grad.norm()) for k, p in vb.named_parameters() if p.grad is not None}
    gslice = {k: p.grad.flatten()[:16].clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor(0.37), cond_token_ids=None, cond=x1, cond_drop_prob=0.0)
    torch.save(dict(loss=loss.detach(), grad_norms=gnorm, grad_slices=gslice, pred_slice=pred[:, :8, :32].clone(),
                    pred_norm=float(pred.norm()), x0_check=x0[0, 0, :4].clone(), times=times, frac=frac, rand=rand),
               os.path.join(HERE, "cfg1.pt"))
    print("cfg1: loss", float(loss))


def gen_cfg4(ref):
    """BASELINE config 4/5 architecture: dim 512, depth 12, heads 16 at B=2, N=1024 (the reference's CPU path at B=8 is the same
    per-sample computation: samples never interact).  Weights by oracle.restate.init_state_dict(seed=4) (410 MB: not
    committed); stored: the loss, gradient norms/slices, one eval prediction slice, and a 4-interval midpoint sample slice."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    state = restate.init_state_dict(cfg, seed=4)
    vb, wrapper = build_reference(ref, cfg, state=state)
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(40))
    x0, times, frac, rand = replay_draws(x1, seed=41)
    torch.manual_seed(41)
    loss = wrapper(x1)
    loss.backward()
    gnorm = {k: float(p.grad.norm()) for k, p in vb.named_parameters() if p.grad is not None}
    gslice = {k: p.grad.flatten()[:16].clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor(0.37), cond_token_ids=None, cond=x1, cond_drop_prob=0.0)
    torch.manual_seed(42)
    y0 = torch.randn_like(x1)
    torch.manual_seed(42)
    s5 = wrapper.sample(cond=x1, steps=5)  # 4 midpoint intervals = 8 function evaluations
    torch.save(dict(loss=loss.detach(), grad_norms=gnorm, grad_slices=gslice, pred_slice=pred[:, :8, :32].clone(),
                    pred_norm=float(pred.norm()), pred_rows=pred[:, 500:504, :].clone(), x0_check=x0[0, 0, :4].clone(),
                    y0_check=y0[0, 0, :4].clone(), times=times, frac=frac, rand=rand,
                    sample5_slice=s5[:, :8, :32].clone(), sample5_rows=s5[:, 500:504, :].clone(), sample5_norm=float(s5.norm())),
               os.path.join(HERE, "cfg4.pt"))
    print("cfg4: loss", float(loss), "pred norm", float(pred.norm()), "sample norm", float(s5.norm()))


def gen_cfg4_wc(ref):
    """cfg4 (dim 512, DEPTH 12, heads 16, B=2, N=1024) with the qk-norm gammas scaled by 0.25: attention logits of std ~5 instead
    of ~80.  At random init with std-80 logits the depth-12 loss is ill-conditioned (tools/precision_ablation.py: rounding ANY one
    operand class to fp16 moves it by 0.6e-3 .. 8e-3, and the fp32 restatement differs from the reference by 5e-5); this variant
    is the well-posed depth-12 parity check: loss within 1e-3, predictions and a 4-interval sample tight."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    state = restate.init_state_dict(cfg, seed=4)
    for k in state:
        if k.endswith("q_norm.gamma") or k.endswith("k_norm.gamma"):
            state[k] = state[k] * 0.25
    vb, wrapper = build_reference(ref, cfg, state=state)
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(40))
    x0, times, frac, rand = replay_draws(x1, seed=41)
    torch.manual_seed(41)
    loss = wrapper(x1)
    loss.backward()
    gnorm = {k: float(p.grad.norm()) for k, p in vb.named_parameters() if p.grad is not None}
    gslice = {k: p.grad.flatten()[:16].clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor(0.37), cond_token_ids=None, cond=x1, cond_drop_prob=0.0)
    torch.manual_seed(42)
    y0 = torch.randn_like(x1)
    torch.manual_seed(42)
    s5 = wrapper.sample(cond=x1, steps=5)
    torch.save(dict(loss=loss.detach(), grad_norms=gnorm, grad_slices=gslice, pred_norm=float(pred.norm()),
                    pred_rows=pred[:, 500:504, :].clone(), x0_check=x0[0, 0, :4].clone(), y0_check=y0[0, 0, :4].clone(),
                    times=times, frac=frac, rand=rand, sample5_rows=s5[:, 500:504, :].clone(), sample5_norm=float(s5.norm())),
               os.path.join(HERE, "cfg4_wc.pt"))
    print("cfg4_wc: loss", float(loss), "pred norm", float(pred.norm()), "sample norm", float(s5.norm()))


def gen_cfg5_wc(ref):
    """BASELINE config 5 on the CPU reference: cfm_wrapper.sample with 64 midpoint intervals (steps = 65 -> 128 function
    evaluations) of the dim-512 / depth-12 / heads-16 network, B = 2 of the 8 (samples never interact), on the well-conditioned
    weights of cfg4_wc (qk-norm gammas x 0.25).  ~1 minute of CPU.  Stored: rows of the final sample and its norm."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    state = restate.init_state_dict(cfg, seed=4)
    for k in state:
        if k.endswith("q_norm.gamma") or k.endswith("k_norm.gamma"):
            state[k] = state[k] * 0.25
    vb, wrapper = build_reference(ref, cfg, state=state)
    vb.eval()
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(40))
    torch.manual_seed(42)
    y0 = torch.randn_like(x1)
    torch.manual_seed(42)
    import time
    t0 = time.time()
    s65 = wrapper.sample(cond=x1, steps=65)
    print("cfg5_wc: 64-interval CPU sample took %.1f s" % (time.time() - t0))
    torch.save(dict(y0_check=y0[0, 0, :4].clone(), sample65_rows=s65[:, 500:516, :].clone(), sample65_first=s65[:, :4, :].clone(),
                    sample65_norm=float(s65.norm()), sample65_absmax=float(s65.abs().max())),
               os.path.join(HERE, "cfg5_wc.pt"))
    print("cfg5_wc: sample norm", float(s65.norm()))


def gen_cfg4_wc_train(ref):
    """Four optimizer steps of the reference on the config-4 architecture (dim 512, depth 12, heads 16, B = 2 x 1024 frames,
    well-conditioned weights of cfg4_wc): the unmodified reference's loss, then clip_grad_norm_(0.5) and torch.optim.Adam(lr 3e-4,
    betas (0.9, 0.99)) exactly as VoiceBoxTrainer.train_step does (trainer.py:258-278, optimizer.py:10-35; no warm-up scheduler here).
    Stored: the per-step losses, the draws of every step, and parameter-update norms of a few tensors after the four steps."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    state = restate.init_state_dict(cfg, seed=4)
    for k in state:
        if k.endswith("q_norm.gamma") or k.endswith("k_norm.gamma"):
            state[k] = state[k] * 0.25
    vb, wrapper = build_reference(ref, cfg, state=state)
    vb.train()
    params = [p for p in vb.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=3e-4, betas=(0.9, 0.99))
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(40))
    losses, draws, gnorms = [], [], []
    for step in range(4):
        x0, times, frac, rand = replay_draws(x1, seed=50 + step)
        torch.manual_seed(50 + step)
        loss = wrapper(x1)
        opt.zero_grad()
        loss.backward()
        gnorms.append(float(torch.nn.utils.clip_grad_norm_(params, 0.5)))
        opt.step()
        losses.append(float(loss))
        draws.append(dict(x0_check=x0[0, 0, :4].clone(), times=times, frac=frac, rand=rand))
        print("cfg4_wc_train step", step, "loss", float(loss), "grad norm", gnorms[-1])
    named = dict(vb.named_parameters())
    keys = ["to_pred.weight", "transformer.layers.0.3.to_qkv.weight", "transformer.layers.11.5.3.weight", "transformer.layers.5.2.to_gamma.weight",
            "transformer.layers.7.3.to_out.weight", "to_embed.weight"]
    upd = {k: float((named[k].detach() - state[k]).norm()) for k in keys}
    torch.save(dict(losses=losses, draws=draws, grad_norms=gnorms, update_norms=upd), os.path.join(HERE, "cfg4_wc_train.pt"))


def gen_small_wc(ref):
    """A WELL-CONDITIONED variant of `small` for the sampler: the qk-norm gammas are scaled by 0.25, so the attention logits
    10*q.k have std ~5 instead of ~80 (a trained checkpoint's regime; at std 80 the softmax is one-hot and the flow field
    is chaotic in its input).  The solver + hipGraph replay must hold a tight tolerance here."""
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    vb, wrapper = build_reference(ref, cfg, seed=0)
    g = torch.Generator().manual_seed(124)
    with torch.no_grad():
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("final_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
            if name.endswith("q_norm.gamma") or name.endswith("k_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
                prm.mul_(0.25)
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    b, n = 2, 40
    x1 = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(70))
    x0, times, frac, rand = replay_draws(x1, seed=97)
    torch.manual_seed(97)
    loss = wrapper(x1)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    cond = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(80))
    with torch.no_grad():
        pred = vb(x1, times=torch.tensor([0.25, 0.8]), cond_token_ids=None, cond=cond, cond_drop_prob=0.0)
        # logit statistics of the first layer, for the record
    out = dict(cfg=dict(dim=64, depth=2, heads=2, dim_head=64), state=state, x1=x1, x0=x0, times=times, frac=frac, rand=rand,
               loss=loss.detach(), grads=grads, cond=cond, eval_times=torch.tensor([0.25, 0.8]), pred=pred)
    for steps in (3, 5, 9, 17):
        torch.manual_seed(30)
        y0 = torch.randn_like(cond)
        torch.manual_seed(30)
        out[f"sample{steps}"] = wrapper.sample(cond=cond, steps=steps)
    out["y0"] = y0
    torch.save(out, os.path.join(HERE, "small_wc.pt"))
    print("small_wc: loss", float(loss))


def gen_small_dropout(ref):
    """Training-time dropout (attn_dropout = 0.1 on the attention probabilities, attend.py:131; ff_dropout = 0.2 between GEGLU and the
    output projection, voicebox_pytorch.py:346) on the well-conditioned small model: one training step of the UNMODIFIED reference
    with forward hooks on its nn.Dropout modules recording which entries survived.  Pins WHERE the restatement applies the masks
    and how they scale; the product path draws its own (Philox) masks and is compared through the restatement."""
    from torch import nn
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    pa, pf = 0.1, 0.2
    torch.manual_seed(0)
    vb = ref.VoiceBox(dim=cfg.dim, num_cond_tokens=500, depth=cfg.depth, dim_head=cfg.dim_head, heads=cfg.heads,
                      condition_on_text=False, num_register_tokens=cfg.num_register_tokens, attn_dropout=pa, ff_dropout=pf)
    wrapper = ref.ConditionalFlowMatcherWrapper(voicebox=vb)
    g = torch.Generator().manual_seed(124)
    with torch.no_grad():
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("final_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
            if name.endswith("q_norm.gamma") or name.endswith("k_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
                prm.mul_(0.25)
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    keep = {"attn": {}, "ff": {}}

    def hook(kind, layer):
        def fn(mod, inp, out):
            keep[kind][layer] = ((out != 0) | (inp[0] == 0)).detach().clone()
        return fn

    n_hooks = 0
    for name, mod in vb.named_modules():
        if isinstance(mod, nn.Dropout) and name.startswith("transformer.layers."):
            layer = int(name.split(".")[2])
            mod.register_forward_hook(hook("attn" if name.endswith("attn_dropout") else "ff", layer))
            n_hooks += 1
    assert n_hooks == 2 * cfg.depth, n_hooks
    b, n = 2, 40
    x1 = torch.randn(b, n, cfg.dim, generator=torch.Generator().manual_seed(71))
    x0, times, frac, rand = replay_draws(x1, seed=96)
    torch.manual_seed(96)
    loss = wrapper(x1)  # wrapper.forward puts the voicebox in train() mode (:1414)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    assert sorted(keep["attn"]) == sorted(keep["ff"]) == list(range(cfg.depth))
    torch.save(dict(cfg=dict(dim=64, depth=2, heads=2, dim_head=64), attn_dropout=pa, ff_dropout=pf, state=state, x1=x1, x0=x0,
                    times=times, frac=frac, rand=rand, keep_attn=keep["attn"], keep_ff=keep["ff"], loss=loss.detach(), grads=grads),
               os.path.join(HERE, "small_dropout.pt"))
    print("small_dropout: loss", float(loss), "kept attn", float(keep["attn"][0].float().mean()), "ff", float(keep["ff"][0].float().mean()))


def gen_small_dimin(ref):
    """dim_in != dim (voicebox_pytorch.py:884,905,938,964): 80-wide data (mel bins) into a dim-64 model -- to_embed is
    Linear(160, 64), to_pred Linear(64, 80), x / cond / target / prediction / ODE state are 80 wide.  Well-conditioned weights;
    loss, every gradient, an eval prediction and a 5-point sample of the unmodified reference."""
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    torch.manual_seed(0)
    vb = ref.VoiceBox(dim=64, dim_in=80, num_cond_tokens=500, depth=2, dim_head=64, heads=2, condition_on_text=False,
                      num_register_tokens=cfg.num_register_tokens)
    wrapper = ref.ConditionalFlowMatcherWrapper(voicebox=vb)
    g = torch.Generator().manual_seed(125)
    with torch.no_grad():
        for name, prm in vb.named_parameters():
            if ".to_gamma." in name or ".to_beta." in name:
                prm.add_(torch.randn(prm.shape, generator=g) * 0.05)
            if name.endswith("final_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
            if name.endswith("q_norm.gamma") or name.endswith("k_norm.gamma"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
                prm.mul_(0.25)
    state = {k: v.detach().clone() for k, v in vb.state_dict().items()}
    assert state["to_embed.weight"].shape == (64, 160) and state["to_pred.weight"].shape == (80, 64) and state["null_cond"].shape == (80,)
    b, n = 2, 40
    x1 = torch.randn(b, n, 80, generator=torch.Generator().manual_seed(72))
    x0, times, frac, rand = replay_draws(x1, seed=95)
    torch.manual_seed(95)
    loss = wrapper(x1)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in vb.named_parameters() if p.grad is not None}
    vb.eval()
    cond = torch.randn(b, n, 80, generator=torch.Generator().manual_seed(81))
    tt = torch.tensor([0.25, 0.8])
    with torch.no_grad():
        pred = vb(x1, times=tt, cond_token_ids=None, cond=cond, cond_drop_prob=0.0)
    torch.manual_seed(31)
    y0 = torch.randn_like(cond)
    torch.manual_seed(31)
    s5 = wrapper.sample(cond=cond, steps=5)
    torch.save(dict(cfg=dict(dim=64, depth=2, heads=2, dim_head=64), dim_in=80, state=state, x1=x1, x0=x0, times=times, frac=frac,
                    rand=rand, loss=loss.detach(), grads=grads, cond=cond, eval_times=tt, pred=pred, y0=y0, sample5=s5),
               os.path.join(HERE, "small_dimin.pt"))
    print("small_dimin: loss", float(loss), "pred", tuple(pred.shape), "sample", tuple(s5.shape))


def _wc(state):
    for k in state:
        if k.endswith("q_norm.gamma") or k.endswith("k_norm.gamma"):
            state[k] = state[k] * 0.25
    return state


def gen_cfg4_seeds(ref):
    """Round 3: the config-4 architecture (dim 512, depth 12, heads 16, N = 1024) at the REFERENCE'S OWN
    initialisation over several seeds -- five at B = 2 (weights seed s, data seed 100+s, draws seed 200+s for s = 10..14) and one at
    B = 8 (s = 15, BASELINE's batch) -- so that the GPU test can assert the |loss difference| DISTRIBUTION instead of one seed.
    Stored per seed: the loss, every gradient norm, the total gradient norm, the draws."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    out = {}
    for s, b in ((10, 2), (11, 2), (12, 2), (13, 2), (14, 2), (15, 8)):
        state = restate.init_state_dict(cfg, seed=s)
        vb, wrapper = build_reference(ref, cfg, state=state)
        x1 = torch.randn(b, 1024, 512, generator=torch.Generator().manual_seed(100 + s))
        x0, times, frac, rand = replay_draws(x1, seed=200 + s)
        torch.manual_seed(200 + s)
        loss = wrapper(x1)
        loss.backward()
        gn = {k: float(p.grad.norm()) for k, p in vb.named_parameters() if p.grad is not None}
        tot = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in vb.parameters() if p.grad is not None)))
        out[s] = dict(batch=b, loss=loss.detach().clone(), grad_norms=gn, grad_total=tot, x0_check=x0[0, 0, :4].clone(),
                      times=times, frac=frac, rand=rand)
        print("cfg4_seeds", s, "B", b, "loss", float(loss), "grad total", tot, flush=True)
        del vb, wrapper, state
    torch.save(out, os.path.join(HERE, "cfg4_seeds.pt"))


def gen_cfg4_seeds_exact(ref=None):
    """Round 4: how well is the reference's OWN loss defined at its initialisation?  The inputs of cfg4_seeds.pt / cfg3.pt["init"]
    through the restatement (oracle/restate.py, pinned against the reference by tests/test_oracle.py) in fp64 -- the exact value of
    the loss -- and in fp32 (the same mathematics in a different operation order).  Stored per case: golden (the unmodified
    reference's fp32 loss), exact, fp32_restatement.  Finding: |exact - reference| is up to 1.3e-3 (seed 12; 1.06e-3 at BASELINE's
    B = 8) and fp32 re-orderings scatter +-1.7e-3 around the exact value: at this initialisation (logit std ~80, a chaotic 12-layer
    map) the reference's loss is defined to ~1e-3, so "within 1e-3 of the reference" is not a property any implementation can have
    on every seed.  tools/reference_noise.py adds the reference's own thread-count spread (3e-5)."""
    out = {}
    cases = [(f"cfg4_seed{s}", restate.Cfg(dim=512, depth=12, heads=16, dim_head=64), s, b, 100 + s, 200 + s, None)
             for s, b in ((10, 2), (11, 2), (12, 2), (13, 2), (14, 2), (15, 8))]
    cases.append(("cfg3_init", restate.Cfg(dim=1024, depth=12, heads=16, dim_head=64), 3, 2, 30, 31, "cfg3"))
    g4 = torch.load(os.path.join(HERE, "cfg4_seeds.pt"), map_location="cpu", weights_only=False)
    g3 = torch.load(os.path.join(HERE, "cfg3.pt"), map_location="cpu", weights_only=False)
    for name, cfg, ws, b, ds, rs, which in cases:
        state = restate.init_state_dict(cfg, seed=ws)
        x1 = torch.randn(b, 1024, cfg.dim, generator=torch.Generator().manual_seed(ds))
        x0, times, frac, rand = replay_draws(x1, seed=rs)
        gold = float((g3["init"] if which else g4[ws])["loss"])
        with torch.no_grad():
            l32 = float(restate.cfm_loss(state, cfg, x1, x0, times, frac, rand))
            p64 = {k: v.double() for k, v in state.items()}
            l64 = float(restate.cfm_loss(p64, cfg, x1.double(), x0.double(), times.double(), frac, rand))
        out[name] = dict(golden=gold, exact=l64, fp32_restatement=l32)
        print(f"{name}: reference {gold:.7f}  exact (fp64) {l64:.7f} ({l64 - gold:+.2e})  fp32 restatement {l32:.7f} ({l32 - gold:+.2e})", flush=True)
    torch.save(out, os.path.join(HERE, "cfg4_seeds_exact.pt"))


def gen_attend_mask4d(ref):
    """Attend.forward (attend.py:100-137) from the unmodified reference with a 4-D key-padding mask (b, 1, 1, j) and with the batch-1
    broadcast form (1, 1, 1, j): outputs and input gradients at the model's logit scale (scale 10 on |q| = |k| = 8)."""
    import importlib

    attend_mod = importlib.import_module("voicebox_pytorch.attend")
    att = attend_mod.Attend(scale=10.0)
    gen = torch.Generator().manual_seed(77)
    q = torch.randn(2, 2, 90, 64, generator=gen)
    k = torch.randn(2, 2, 90, 64, generator=gen)
    q, k = q / q.norm(dim=-1, keepdim=True) * 8, k / k.norm(dim=-1, keepdim=True) * 8
    v = torch.randn(2, 2, 90, 64, generator=gen)
    # the goldens are taken on fp16-representable inputs (what the kernels see)
    q, k, v = q.half().float(), k.half().float(), v.half().float()
    out = {"q": q, "k": k, "v": v}
    m2 = torch.ones(2, 90, dtype=torch.bool)
    m2[1, 60:] = False
    m2[0, 3:7] = False
    for name, mask in (("b11j", m2[:, None, None, :]), ("111j", m2[1:2, None, None, :])):
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
        o = att(qr, kr, vr, mask=mask)
        w = torch.randn(o.shape, generator=torch.Generator().manual_seed(5))
        (o * w).sum().backward()
        out[name] = dict(mask=mask.clone(), out=o.detach(), w=w, dq=qr.grad.clone(), dk=kr.grad.clone(), dv=vr.grad.clone())
    torch.save(out, os.path.join(HERE, "attend_mask4d.pt"))
    print("attend_mask4d: out norm", float(out["b11j"]["out"].norm()))


def gen_cfg3(ref):
    """Round 3: BASELINE config 3 -- dim 1024, heads 16, depth 12 -- at B = 2, N = 1024 from the unmodified
    reference, once at the reference's initialisation and once well-conditioned (qk-norm gammas x 0.25).  Weights by
    oracle.restate.init_state_dict(cfg, seed=3) (1.3 GB: not committed).  Stored: loss, every gradient norm + 16-element slices,
    rows of one eval prediction."""
    cfg = restate.Cfg(dim=1024, depth=12, heads=16, dim_head=64)
    out = {}
    for name in ("init", "wc"):
        state = restate.init_state_dict(cfg, seed=3)
        if name == "wc":
            _wc(state)
        vb, wrapper = build_reference(ref, cfg, state=state)
        x1 = torch.randn(2, 1024, 1024, generator=torch.Generator().manual_seed(30))
        x0, times, frac, rand = replay_draws(x1, seed=31)
        torch.manual_seed(31)
        loss = wrapper(x1)
        loss.backward()
        gn = {k: float(p.grad.norm()) for k, p in vb.named_parameters() if p.grad is not None}
        gs = {k: p.grad.flatten()[:16].clone() for k, p in vb.named_parameters() if p.grad is not None}
        vb.eval()
        with torch.no_grad():
            pred = vb(x1, times=torch.tensor(0.37), cond_token_ids=None, cond=x1, cond_drop_prob=0.0)
        out[name] = dict(loss=loss.detach().clone(), grad_norms=gn, grad_slices=gs, pred_norm=float(pred.norm()),
                         pred_rows=pred[:, 500:504, :].clone(), x0_check=x0[0, 0, :4].clone(), times=times, frac=frac, rand=rand)
        print("cfg3", name, "loss", float(loss), "pred norm", float(pred.norm()), flush=True)
        del vb, wrapper, state
    torch.save(out, os.path.join(HERE, "cfg3.pt"))


def gen_cfg5_wc_b8(ref):
    """Round 3: BASELINE config 5 at its own batch -- cfm_wrapper.sample(cond = (8, 1024, 512), steps = 65) = 64
    midpoint intervals = 128 function evaluations of the dim-512 / depth-12 network on the unmodified reference's CPU path, EIGHT
    distinct samples, well-conditioned weights of cfg4_wc.  ~15-20 minutes of CPU.  Stored: 16 rows of every sample + norms."""
    cfg = restate.Cfg(dim=512, depth=12, heads=16, dim_head=64)
    state = _wc(restate.init_state_dict(cfg, seed=4))
    vb, wrapper = build_reference(ref, cfg, state=state)
    vb.eval()
    x1 = torch.randn(8, 1024, 512, generator=torch.Generator().manual_seed(48))
    torch.manual_seed(49)
    y0 = torch.randn_like(x1)
    torch.manual_seed(49)
    import time
    t0 = time.time()
    s65 = wrapper.sample(cond=x1, steps=65)
    print("cfg5_wc_b8: 64-interval CPU sample of 8 took %.1f s" % (time.time() - t0), flush=True)
    torch.save(dict(y0_check=y0[:, 0, :4].clone(), sample65_rows=s65[:, 500:516, :].clone(), sample65_first=s65[:, :4, :].clone(),
                    sample65_norms=s65.flatten(1).norm(dim=1).clone(), sample65_absmax=float(s65.abs().max()),
                    cpu_seconds=time.time() - t0),
               os.path.join(HERE, "cfg5_wc_b8.pt"))
    print("cfg5_wc_b8: sample norms", s65.flatten(1).norm(dim=1).tolist())


def gen_init_stats(ref):
    """Round 5: the reference-initialisation parity statement as a STATISTIC over many seeds instead of bounds fitted
    to six realisations.  Forward only.  config 4 architecture (dim 512, depth 12, heads 16, N = 1024): 16 seeds at B = 2 (weights seed
    s, data seed 100 + s, draws seed 200 + s for s = 20..35) ; config 3 (dim 1024): 6 seeds at B = 2 (s = 40..45).  Per seed: the
    unmodified reference's fp32 loss, the exact (fp64 restatement) loss and the fp32 restatement's loss (= what another correct fp32
    implementation gets), plus the draws.  tests/test_model_gpu.py asserts mean |difference| and RMS over the seeds."""
    out = {}
    for tag, cfg, seeds in (("cfg4", restate.Cfg(dim=512, depth=12, heads=16, dim_head=64), range(20, 36)),
                            ("cfg3", restate.Cfg(dim=1024, depth=12, heads=16, dim_head=64), range(40, 46))):
        for s in seeds:
            state = restate.init_state_dict(cfg, seed=s)
            vb, wrapper = build_reference(ref, cfg, state=state)
            x1 = torch.randn(2, 1024, cfg.dim, generator=torch.Generator().manual_seed(100 + s))
            x0, times, frac, rand = replay_draws(x1, seed=200 + s)
            torch.manual_seed(200 + s)
            with torch.no_grad():
                gold = float(wrapper(x1))
                l32 = float(restate.cfm_loss(state, cfg, x1, x0, times, frac, rand))
                p64 = {k: v.double() for k, v in state.items()}
                l64 = float(restate.cfm_loss(p64, cfg, x1.double(), x0.double(), times.double(), frac, rand))
            out[(tag, s)] = dict(loss=gold, exact=l64, fp32_restatement=l32, x0_check=x0[0, 0, :4].clone(), times=times, frac=frac, rand=rand)
            print(f"init_stats {tag} seed {s}: reference {gold:.7f}  exact {l64:.7f} ({l64 - gold:+.2e})  fp32 restatement {l32:.7f} ({l32 - gold:+.2e})", flush=True)
            del vb, wrapper, state, p64
    torch.save(out, os.path.join(HERE, "init_stats.pt"))


if __name__ == "__main__":
    ref = ref_loader.load_reference()
    which = sys.argv[1:] or ["masks", "rotary", "small", "small_gateloop", "small_text", "transformer", "duration", "cfg1", "cfg4",
                             "small_wc", "cfg4_wc", "cfg5_wc", "cfg4_wc_train", "cfg4_seeds", "cfg3", "cfg5_wc_b8", "small_dropout", "small_dimin", "cfg4_seeds_exact", "attend_mask4d", "transformer_unet", "init_stats"]
    for w in which:
        {"masks": gen_masks, "rotary": gen_rotary, "small": gen_small, "small_gateloop": gen_small_gateloop, "small_text": gen_small_text,
         "transformer": gen_transformer, "duration": gen_duration, "cfg1": gen_cfg1, "cfg4": gen_cfg4, "small_wc": gen_small_wc, "cfg4_wc": gen_cfg4_wc,
         "cfg5_wc": gen_cfg5_wc, "cfg4_wc_train": gen_cfg4_wc_train, "cfg4_seeds": gen_cfg4_seeds, "cfg3": gen_cfg3,
         "cfg5_wc_b8": gen_cfg5_wc_b8, "small_dropout": gen_small_dropout, "small_dimin": gen_small_dimin,
         "cfg4_seeds_exact": gen_cfg4_seeds_exact, "attend_mask4d": gen_attend_mask4d, "transformer_unet": gen_transformer_unet,
         "init_stats": gen_init_stats}[w](ref)
