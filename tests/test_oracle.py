"""CPU: the oracle restatement (oracle/restate.py) against the golden vectors the UNMODIFIED
reference produced (tests/golden/make_golden.py), plus -- when /root/reference is present --
directly against the reference."""
import pytest
import torch

from oracle import restate, ref_loader


def _cfg(d):
    return restate.Cfg(**d)


def test_masks_bit_exact(golden):
    g = golden("masks")
    for n, expect in g["cases"].items():
        got = restate.frac_lengths_mask(n, g["frac"], g["rand"])
        assert torch.equal(got, expect), n
    assert torch.equal(restate.span_mask(8, g["start"], g["end"]), g["start_end_8"])
    assert torch.equal(restate.frac_lengths_mask(1024, g["frac"], g["frac_helper_rand"]), g["frac_helper_1024"])
    # SURVEY 3.4 #5 known answers
    m = restate.frac_lengths_mask(1024, torch.tensor([0.7, 0.85, 0.9999, 1.0]), torch.zeros(4))
    assert m.sum(-1).tolist() == [716, 870, 1023, 1024]
    assert restate.span_mask(8, torch.tensor([2.9]), torch.tensor([5.9]))[0].int().tolist() == [0, 0, 1, 1, 1, 0, 0, 0]


def test_rotary_bit_exact(golden):
    g = golden("rotary")
    fr = restate.rotary_freqs(g["positions"], 64, 50000.0)
    assert torch.equal(fr, g["freqs"])
    assert torch.equal(restate.apply_rotary(fr, g["t"]), g["rotated"])


def test_small_model_loss_and_grads(golden):
    g = golden("small")
    cfg = _cfg(g["cfg"])
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
    loss = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, ref in g["grads"].items():
        got = p[k].grad
        assert got is not None, k
        assert torch.allclose(got, ref, rtol=2e-3, atol=2e-6), (k, float((got - ref).abs().max()))
    # padded batch
    p2 = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
    loss_m = restate.cfm_loss(p2, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"], mask=g["mask"])
    assert abs(float(loss_m) - float(g["loss_masked"])) < 1e-5
    loss_m.backward()
    for k, ref in g["grads_masked"].items():
        assert torch.allclose(p2[k].grad, ref, rtol=2e-3, atol=2e-6), k


def test_small_model_with_dropout_masks_of_the_reference(golden):
    """attend.py:131 / voicebox_pytorch.py:346: the restatement with the keep masks the reference's own nn.Dropout modules drew
    (recorded by forward hooks, tests/golden/make_golden.py::gen_small_dropout) reproduces the reference's loss and gradients --
    i.e. the masks act where and how nn.Dropout acts (on the softmax output, on the GEGLU output; survivors / (1 - p))."""
    g = golden("small_dropout")
    cfg = _cfg(g["cfg"])
    pa, pf = g["attn_dropout"], g["ff_dropout"]
    attn = {l: k.float() / (1 - pa) for l, k in g["keep_attn"].items()}
    ff = {l: k.float() / (1 - pf) for l, k in g["keep_ff"].items()}
    assert 0.85 < float(g["keep_attn"][0].float().mean()) < 0.95 and 0.75 < float(g["keep_ff"][0].float().mean()) < 0.85
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
    with restate.dropout_multipliers(attn=attn, ff=ff):
        loss = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, ref in g["grads"].items():
        assert torch.allclose(p[k].grad, ref, rtol=2e-3, atol=2e-6), (k, float((p[k].grad - ref).abs().max()))
    # and without the masks the loss differs (the masks matter at this tolerance)
    loss0 = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"])
    assert abs(float(loss0) - float(g["loss"])) > 1e-3


def test_small_model_with_dim_in_other_than_dim(golden):
    """dim_in != dim (voicebox_pytorch.py:884,905,938,964): the restatement follows the weight shapes, so 80-wide data through a
    dim-64 model needs no special case -- pinned on the unmodified reference's loss, gradients, prediction and 5-point sample."""
    g = golden("small_dimin")
    cfg = _cfg(g["cfg"])
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
    loss = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, ref in g["grads"].items():
        assert torch.allclose(p[k].grad, ref, rtol=2e-3, atol=2e-6), (k, float((p[k].grad - ref).abs().max()))
    with torch.no_grad():
        ones = torch.ones(g["x1"].shape[:2], dtype=torch.bool)
        pred = restate.voicebox_forward(g["state"], cfg, g["x1"], g["eval_times"], g["cond"], ones)
        assert pred.shape[-1] == 80 and torch.allclose(pred, g["pred"], rtol=1e-4, atol=1e-5)
        s5 = restate.sample_midpoint(g["state"], cfg, g["y0"], 5, cond=g["cond"])
        assert torch.allclose(s5, g["sample5"], rtol=1e-3, atol=1e-4), float((s5 - g["sample5"]).abs().max())


def test_small_model_eval_and_sample(golden):
    g = golden("small")
    cfg = _cfg(g["cfg"])
    p = g["state"]
    with torch.no_grad():
        pred = restate.voicebox_forward(p, cfg, g["x1"], g["eval_times"], g["cond"], g["cond_mask"])
        assert torch.allclose(pred, g["pred"], rtol=1e-4, atol=1e-5)
        ones = torch.ones(g["x1"].shape[:2], dtype=torch.bool)
        pred_s = restate.voicebox_forward(p, cfg, g["x1"], torch.tensor(0.5), g["cond"], ones)
        assert torch.allclose(pred_s, g["pred_scalar_t"], rtol=1e-4, atol=1e-5)
        for steps, key in ((3, "sample3"), (5, "sample5")):
            s = restate.sample_midpoint(p, cfg, g["y0"], steps)
            assert torch.allclose(s, g[key], rtol=1e-4, atol=1e-5), key


def test_small_gateloop_model(golden):
    """use_gateloop_layers=True: restated stack vs the reference module tree run around the restated third-party
    layer (its own arithmetic is parity-unpinned, see oracle/restate.py:gateloop)."""
    g = golden("small_gateloop")
    cfg = _cfg(g["cfg"])
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
    loss = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, ref in g["grads"].items():
        got = p[k].grad
        assert got is not None, k
        assert float((got - ref).norm() / ref.norm().clamp(min=1e-20)) < 2e-3, k
    # an independent formulation of the scan (closed form through cumulative log-gates) agrees with the loop
    torch.manual_seed(0)
    q, kv, a = torch.randn(3, 2, 37, 8, dtype=torch.float64).unbind(0)
    a = a.sigmoid()
    logc = a.log().cumsum(1)
    closed = q * (torch.exp(logc) * (kv * torch.exp(-logc)).cumsum(1))
    h = torch.zeros(2, 8, dtype=torch.float64)
    loop = []
    for t in range(37):
        h = a[:, t] * h + kv[:, t]
        loop.append(q[:, t] * h)
    assert torch.allclose(torch.stack(loop, 1), closed, rtol=1e-9, atol=1e-9)


def test_text_conditioned_model(golden):
    """condition_on_text=True: embedding gather + bilinear token->frame resize + classifier-free-guidance drop, eval and guided
    sampling, against the unmodified reference."""
    g = golden("small_text")
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    for ids_key, loss_key, grads_key, drop in (("ids", "loss", "grads", g["drop"]), ("ids_n", "loss_n", "grads_n", None)):
        p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in g["state"].items()}
        loss = restate.cfm_loss(p, cfg, g["x1"], g["x0"], g["times"], g["frac"], g["rand"], cond_token_ids=g[ids_key],
                                cond_drop_mask=drop)
        assert abs(float(loss) - float(g[loss_key])) < 1e-5, (ids_key, float(loss), float(g[loss_key]))
        loss.backward()
        for k, ref in g[grads_key].items():
            assert p[k].grad is not None, k
            assert float((p[k].grad - ref).norm() / ref.norm().clamp(min=1e-20)) < 2e-3, (ids_key, k)
    st = g["state"]
    b, n = g["x1"].shape[:2]
    ones = torch.ones(b, n, dtype=torch.bool)  # eval with cond_mask None (:1028-1029)
    with torch.no_grad():
        pred = restate.voicebox_forward(st, cfg, g["x1"], torch.tensor(0.4), g["cond"], ones, cond_token_ids=g["ids"])
        assert float((pred - g["pred"]).norm() / g["pred"].norm()) < 1e-5
        pc = restate.forward_with_cond_scale(st, cfg, g["x1"], torch.tensor(0.4), g["cond"], ones, g["ids"], 1.7)
        assert float((pc - g["pred_cfg"]).norm() / g["pred_cfg"].norm()) < 1e-5
        s3 = restate.sample_midpoint(st, cfg, g["y0"], 3, cond=g["cond"], cond_token_ids=g["ids_n"], cond_scale=1.3)
        assert float((s3 - g["sample3"]).norm() / g["sample3"].norm()) < 1e-4


def test_standalone_transformer(golden):
    """restate.transformer (adaptive + plain RMSNorm, with/without registers and qk-norm) vs the reference's Transformer.forward."""
    g = golden("transformer")
    for name, c in g.items():
        kw = c["kw"]
        cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64, num_register_tokens=kw["num_register_tokens"],
                          qk_norm=kw["attn_qk_norm"])
        p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in c["state"].items()}
        x = c["x"].clone().requires_grad_(True)
        cond = c["cond"].clone().requires_grad_(True) if c["cond"] is not None else None
        y = restate.transformer(x, p, cfg, mask=c["mask"], cond=cond, pre="")
        assert float((y - c["y"]).norm() / c["y"].norm()) < 1e-5, name
        (y * c["dout"]).sum().backward()
        assert float((x.grad - c["dx"]).norm() / c["dx"].norm()) < 2e-3, name
        if cond is not None:
            assert float((cond.grad - c["dcond"]).norm() / c["dcond"].norm()) < 2e-3, name
        for k, ref in c["grads"].items():
            assert float((p[k].grad - ref).norm() / ref.norm().clamp(min=1e-20)) < 2e-3, (name, k)


def test_standalone_transformer_unet(golden):
    """restate.transformer with the u-net skip combiners (voicebox_pytorch.py:453-463; depth 4, custom and default
    skip_connect_scale, with GateLoop layers) vs the reference's Transformer(use_unet_skip_connection=True)."""
    g = golden("transformer_unet")
    for name, c in g.items():
        kw = c["kw"]
        cfg = restate.Cfg(dim=64, depth=4, heads=2, dim_head=64, num_register_tokens=kw["num_register_tokens"],
                          qk_norm=kw["attn_qk_norm"], use_gateloop=bool(kw.get("use_gateloop_layers")))
        cfg.skip_connect_scale = kw.get("skip_connect_scale") or 2 ** -0.5
        p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in c["state"].items()}
        x = c["x"].clone().requires_grad_(True)
        cond = c["cond"].clone().requires_grad_(True) if c["cond"] is not None else None
        y = restate.transformer(x, p, cfg, mask=c["mask"], cond=cond, pre="")
        assert float((y - c["y"]).norm() / c["y"].norm()) < 1e-5, name
        (y * c["dout"]).sum().backward()
        assert float((x.grad - c["dx"]).norm() / c["dx"].norm()) < 2e-3, name
        if cond is not None:
            assert float((cond.grad - c["dcond"]).norm() / c["dcond"].norm()) < 2e-3, name
        for k, ref in c["grads"].items():  # stored as bf16
            assert float((p[k].grad - ref.float()).norm() / ref.float().norm().clamp(min=1e-20)) < 1e-2, (name, k)


def test_duration_predictor_inference(golden):
    """restate.duration_predictor_* vs the reference's DurationPredictor in eval mode (durations, CFG mix, aligned ids)."""
    g = golden("duration")
    for name, c in g.items():
        cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64, num_register_tokens=0, qk_norm=c["kw"]["attn_qk_norm"])
        p = c["state"]
        with torch.no_grad():
            d1 = restate.duration_predictor_forward(p, cfg, c["cond"], c["ids"], c["cond_mask"])
            dn = restate.duration_predictor_forward(p, cfg, c["cond"], c["ids"], c["cond_mask"],
                                                    cond_drop_mask=torch.ones(3, dtype=torch.bool))
            d3 = restate.duration_predictor_with_cond_scale(p, cfg, c["cond"], c["ids"], c["cond_mask"], cond_scale=3.0)
        for got, key in ((d1, "d1"), (dn, "d_null"), (d3, "d3")):
            assert float((got - c[key]).abs().max()) < 1e-4, (name, key)
        assert float((c["d1"] - c["d_null"]).abs().max()) > 1e-2, name  # the null condition changes the answer
        # index logic: bit-exact on the reference's own durations
        assert torch.equal(restate.align_phoneme_ids_with_durations(c["ids"], c["d3"]), c["aligned"]), name
    # known answers of the repeat mask (a padded repeat_interleave)
    m = restate.generate_mask_from_repeats(torch.tensor([[2, 1, 3], [1, 1, 1]]))
    assert m.shape == (2, 3, 6)
    assert m[0].int().tolist() == [[1, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]]
    assert m[1].int().tolist() == [[1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0]]
    al = restate.align_phoneme_ids_with_durations(torch.tensor([[7, 9, 4]]), torch.tensor([[2.7, 0.2, 3.0]]))
    assert al.tolist() == [[7, 7, 9, 4, 4, 4]]  # clamp(min=1) then int() truncation


def test_cfg1_loss(golden):
    """BASELINE config 1 (dim 512, depth 2, B=2, N=1024) on CPU: restatement vs reference scalars."""
    g = golden("cfg1")
    cfg = restate.Cfg(dim=512, depth=2, heads=16, dim_head=64)
    p = restate.init_state_dict(cfg, seed=0)
    x1 = torch.randn(2, 1024, 512, generator=torch.Generator().manual_seed(0))
    torch.manual_seed(1)
    x0 = torch.randn_like(x1)
    assert torch.equal(x0[0, 0, :4], g["x0_check"])
    with torch.no_grad():
        loss = restate.cfm_loss(p, cfg, x1, x0, g["times"], g["frac"], g["rand"])
    assert abs(float(loss) - float(g["loss"])) < 2e-5


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference sources only exist in the build container")
def test_restatement_vs_live_reference():
    ref = ref_loader.load_reference()
    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=4)
    vb = ref.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    wrapper = ref.ConditionalFlowMatcherWrapper(voicebox=vb)
    x1 = torch.randn(3, 50, 64)
    torch.manual_seed(5)
    ref_loss = wrapper(x1)
    torch.manual_seed(5)
    x0 = torch.randn_like(x1)
    times = torch.rand((3,))
    frac = torch.zeros((3,)).float().uniform_(0.7, 1.0)
    rand = torch.zeros_like(frac).float().uniform_(0, 1)
    got = restate.cfm_loss(state, cfg, x1, x0, times, frac, rand)
    assert abs(float(got) - float(ref_loss)) < 1e-5


def test_midpoint_restatement_reproduces_the_textbook_rule():
    """torchdiffeq is absent (parity of the solver is unpinned against IT); what CAN be pinned is that the restated fixed-grid
    midpoint integrator is the textbook explicit midpoint rule: for y' = a(t) y with a(t) = c0 + c1 t one step is
    y <- y (1 + h a(t + h/2) (1 + h a(t) / 2)), and the whole grid is the product of those factors."""
    from oracle.ref_loader import odeint_fixed_grid_midpoint

    c0, c1 = -1.3, 0.7
    for steps in (2, 3, 17, 64, 65):
        t = torch.linspace(0, 1, steps, dtype=torch.float64)
        y0 = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
        ys = odeint_fixed_grid_midpoint(lambda tt, y: (c0 + c1 * tt) * y, y0, t)
        fac = 1.0
        for i in range(steps - 1):
            h = float(t[i + 1] - t[i]); ti = float(t[i])
            fac *= 1.0 + h * (c0 + c1 * (ti + h / 2)) * (1.0 + h * (c0 + c1 * ti) / 2)
        assert ys.shape[0] == steps
        assert torch.allclose(ys[-1], y0 * fac, rtol=1e-13, atol=0)
        # second-order convergence towards the exact solution exp(c0 + c1/2)
        exact = y0 * torch.exp(torch.tensor(c0 + c1 / 2, dtype=torch.float64))
        err = float((ys[-1] - exact).abs().max())
        assert err < 2.0 / (steps - 1) ** 2
