"""GPU: the full data-parallel train step (voicebox_pytorch_amd.dp.TrainStep) with world_size 2.  Both ranks share the
single test GPU and exchange gradients over gloo (RCCL refuses two ranks on one device); the code path -- staged backward,
bucketed async all-reduce on a side stream, clip + fused Adam -- is the one bench.py runs over RCCL on 8 GPUs.

Checked on rank 0: the reduced flat gradient equals the mean of the two shards' gradients computed locally one after the
other (HIP kernels are deterministic -> equality up to one fp32 add), and both ranks end with identical parameters."""
import math
import os
import socket
import sys

import zlib

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
dev = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _draws(seed, B, N, D):
    g = torch.Generator().manual_seed(seed)
    return dict(x1=torch.randn(B, N, D, generator=g), x0=torch.randn(B, N, D, generator=g), times=torch.rand(B, generator=g),
                frac_lengths=0.7 + 0.3 * torch.rand(B, generator=g), rand=torch.rand(B, generator=g))


def _worker(rank, world, port, out, grad_mode="allreduce", adaln_grads="auto"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override
    from oracle import restate

    cfg = restate.Cfg(dim=128, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=5 + rank)  # different initial weights: the broadcast must fix that
    vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    vb = vb.to("cuda:0")
    wrapper = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    ts = TrainStep(wrapper, lr=1e-3, max_grad_norm=0.5, bucket_bytes=1 << 16, grad_mode=grad_mode, adaln_grads=adaln_grads)  # small buckets: several async collectives
    B, N, D = 2, 72, 128
    shards = [_draws(100 + r, B, N, D) for r in range(world)]
    mine = shards[rank]
    p_before = ts.fp.flat.clone()
    with rng_override(**{k: v for k, v in mine.items() if k != "x1"}):
        loss = ts.step(mine["x1"].cuda())
    torch.cuda.synchronize()
    g_reduced = ts.gflat.clone() / world
    p_after = ts.fp.flat.clone()
    owned = list(ts._red.owned) if grad_mode == "shard" else None  # shard mode: only the owned chunks hold the sum
    # parameters must be identical on both ranks after the step
    gathered = [torch.zeros_like(p_after) for _ in range(world)]
    dist.all_gather(gathered, p_after)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        # local reference: both shards' gradients with the pre-step weights, no reducer
        vb2 = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False).to("cuda:0")
        fp2 = vb2.flat_params()
        fp2.flat.copy_(p_before)
        w2 = vbx.ConditionalFlowMatcherWrapper(voicebox=vb2)
        acc = torch.zeros_like(p_before)
        for sh in shards:
            vb2.zero_grad(set_to_none=True)
            with rng_override(**{k: v for k, v in sh.items() if k != "x1"}):
                l2 = w2(sh["x1"].cuda())
            l2.backward()
            for slot in fp2.order:
                prm = fp2.slots[slot]
                o = fp2.offsets[slot]
                acc[o:o + prm.numel()] += prm.grad.flatten()
        ref = acc / world
        if owned is None:
            err = float((g_reduced - ref).abs().max())
        else:
            err = max(float((g_reduced[lo:hi] - ref[lo:hi]).abs().max()) for lo, hi in owned)
        scale = float(ref.abs().max())
        out.put((err, scale, same, float(loss), float((p_after - p_before).abs().max()), p_after.cpu().numpy()))  # by value: a shared-memory tensor handle dies with the worker
    dist.barrier()
    dist.destroy_process_group()


def test_last_micro_batch_with_overlapped_exchange_equals_deferred_exchange():
    """accumulate_last_and_apply (the accumulation window's last micro-batch folds the accumulator in stage by stage and starts each
    stage's exchange under its own backward) against accumulate + apply_accumulated (everything exchanged after the backward):
    same parameters after the step up to the rounding of one fused multiply-add per gradient element."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override

    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(2, 72, 128, generator=g) for _ in range(2)]
    draws = [dict(x0=torch.randn(2, 72, 128, generator=g), times=torch.rand(2, generator=g), frac_lengths=0.7 + 0.3 * torch.rand(2, generator=g),
                  rand=torch.rand(2, generator=g)) for _ in range(2)]
    flats = []
    for fused in (False, True):
        torch.manual_seed(1)
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False).to("cuda:0")
        ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb), lr=3e-4, max_grad_norm=0.5)
        with rng_override(**draws[0]):
            ts.accumulate(xs[0].cuda(), 0.5)
        with rng_override(**draws[1]):
            if fused:
                ts.accumulate_last_and_apply(xs[1].cuda(), 0.5)
            else:
                ts.accumulate(xs[1].cuda(), 0.5)
                ts.apply_accumulated()
        torch.cuda.synchronize()
        flats.append((ts.fp.flat.clone(), ts.m.clone()))
    dm = float((flats[0][1] - flats[1][1]).abs().max()) / float(flats[0][1].abs().max())
    assert dm < 1e-5, dm                                       # first moments = (1 - beta1) * clipped gradient: the gradients agree
    assert float((flats[0][0] - flats[1][0]).abs().max()) < 1e-6  # and so do the updated parameters


def test_length_bucketing_pads_with_masked_frames_only():
    """TrainStep(length_bucket=64): a batch of 203 frames runs on the 256-frame engine (padded frames are masked everywhere) and gives
    the loss and the gradients of the exact-length run up to summation order; ragged key-padding masks included."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override

    g = torch.Generator().manual_seed(9)
    B, N = 3, 203
    x = torch.randn(B, N, 128, generator=g)
    draws = dict(x0=torch.randn(B, N, 128, generator=g), times=torch.rand(B, generator=g), frac_lengths=0.7 + 0.3 * torch.rand(B, generator=g),
                 rand=torch.rand(B, generator=g))
    mask = torch.ones(B, N, dtype=torch.bool)
    mask[1, 150:] = False
    res = []
    for lb in (0, 64):
        torch.manual_seed(1)
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False).to("cuda:0")
        with torch.no_grad():
            for name, p in vb.named_parameters():
                if ".to_gamma.weight" in name or ".to_beta." in name:
                    p.normal_(0.0, 0.02, generator=None)
        ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb), lr=3e-4, max_grad_norm=0.5, length_bucket=lb)
        with rng_override(**draws):
            loss = ts._forward_backward(x.cuda(), mask.cuda(), None, on_stage=None)
        torch.cuda.synchronize()
        res.append((float(loss), ts.gflat.clone(), sorted(k[1] for k in vb._engines)))
    assert res[0][2] == [203] and res[1][2] == [256]
    assert abs(res[0][0] - res[1][0]) < 1e-5, (res[0][0], res[1][0])
    rel = float((res[0][1] - res[1][1]).norm() / res[0][1].norm())
    assert rel < 2e-3, rel  # bf16-operand GEMMs over a different number of rows: split-K boundaries and tile tails move


def test_train_step_world2_on_one_gpu():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, same, loss, moved, _ = out.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert same, "ranks diverged"
    assert err <= 1e-6 * max(scale, 1e-6) + 1e-9, (err, scale)
    assert moved > 0 and moved < 2e-3  # Adam moved every weight by at most ~lr
    assert 1.0 < loss < 10.0


def test_train_step_shard_mode_equals_allreduce_mode_world2_on_one_gpu():
    """TrainStep(grad_mode="shard"): reduce-scatter of the buckets, clip + Adam on the owned chunks, all-gather of the parameters -- the
    same step as the all-reduce mode (identical parameters on both ranks, gradients of the owned chunks = the full-batch gradient,
    updated parameters equal to the all-reduce mode's up to the summation order of the gradient norm)."""
    ctx = mp.get_context("spawn")
    res = {}
    for mode in ("allreduce", "shard"):
        out = ctx.Queue()
        port = _free_port()
        # (adaln_grads="materialize" in both: shard mode always materialises the adaLN weight gradients, and this test compares the
        #  two exchanges bit-wise; the factor exchange of the all-reduce mode is test_train_step_world2_on_one_gpu's default)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, out, mode, "materialize")) for r in range(2)]
        for p in procs:
            p.start()
        res[mode] = out.get(timeout=600)
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
    err, scale, same, loss, moved, p_shard = res["shard"]
    assert same, "ranks diverged in shard mode"
    assert err <= 1e-6 * max(scale, 1e-6) + 1e-9, (err, scale)
    import numpy as np

    d = float(np.abs(p_shard - res["allreduce"][5]).max())
    assert d < 2e-7, d  # one Adam step of size ~lr = 1e-3: the clip coefficient differs in the last fp32 bits only


def test_factor_mode_falls_back_to_materialised_gradients_above_batch_64():
    """the one-GPU factor-form gradient norm (vbx_sumsq_adaln_factors: B x B Gram terms) serves local batches up to 64.
    TrainStep's default adaln_grads="auto" must train a batch of 65 (materialised gradients) instead of raising after the backward,
    a batch of 64 still runs in factor form with a scratch sized for its 64 x 64 terms, and an explicit "factors" request says so."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep

    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False).to(dev)
    ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb), lr=1e-3, max_grad_norm=0.5)
    assert ts.adaln_factors_apply() and ts.adaln_factors_apply(batch=64) and not ts.adaln_factors_apply(batch=65)
    g = torch.Generator().manual_seed(5)
    for B in (65, 64, 65):
        p0 = ts.fp.flat.clone()
        loss = ts.step(torch.randn(B, 24, 64, generator=g).to(dev))
        torch.cuda.synchronize()
        assert math.isfinite(float(loss)) and bool(torch.isfinite(ts.fp.flat).all())
        assert float((ts.fp.flat - p0).abs().max()) > 0  # the step moved the parameters
    ts2 = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb), lr=1e-3, max_grad_norm=0.5, adaln_grads="factors")
    with pytest.raises(AssertionError):
        ts2.step(torch.randn(65, 24, 64, generator=g).to(dev))


def test_adaln_factor_mode_equals_materialised_gradients():
    """TrainStep(adaln_grads="factors") on one GPU (round 5): the adaLN projection weight gradients are never written -- the global norm
    takes their sum of squares from the B x B Gram matrices of the factors (vbx_sumsq_adaln_factors + vbx_sumsq_ranges), Adam expands
    dada_l^T . temb on the fly (vbx_adam_adaln_factors) -- against adaln_grads="materialize" (round 4's path) from the same weights and
    draws: same clip coefficient and gradient norm to fp32 rounding, same parameters after each of three steps (the update of every
    tensor, adaLN weights included, to 1e-4 of its norm), the fp16 operand copies of the adaLN weights refreshed (next forward's loss
    equal), every other gradient bit-identical, and the adaLN weight blocks of the gradient buffer left untouched."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override
    from oracle import restate

    cfg = restate.Cfg(dim=128, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=11)
    for k in state:  # the reference zero-initialises the adaLN projections (:264-268): randomise them so that they matter
        if ".to_gamma." in k or ".to_beta." in k:
            state[k] = state[k] + 0.05 * torch.randn(state[k].shape, generator=torch.Generator().manual_seed(zlib.crc32(k.encode()) % 1000))
    d = _draws(321, 3, 72, 128)
    runs = {}
    for mode in ("materialize", "factors"):
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(state, strict=False)
        ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to(dev)), lr=1e-3, max_grad_norm=0.5, adaln_grads=mode)
        assert ts.adaln_factors_apply() == (mode == "factors")
        ts.gflat.fill_(7.0)  # sentinel: factor mode must not touch the adaLN weight blocks
        p0 = ts.fp.flat.clone()
        rec = []
        for _ in range(3):
            with rng_override(**{k: v for k, v in d.items() if k != "x1"}):
                loss = ts.step(d["x1"].to(dev))
            torch.cuda.synchronize()
            rec.append((float(loss), ts.coef.clone().cpu(), ts.fp.flat.clone()))
        runs[mode] = dict(rec=rec, g=ts.gflat.clone(), p0=p0, ranges=ts.adaln_weight_ranges(), fp=ts.fp)
    a, b = runs["materialize"], runs["factors"]
    for step, ((la, ca, pa), (lb, cb, pb)) in enumerate(zip(a["rec"], b["rec"])):
        # step 0: the same forward.  Step 1: a forward at parameters that agree to 1e-4 per tensor.  Step 2: two updates in, at this
        # (reference-style, chaotic) init the trajectories have separated -- 2e-3 .. 9e-3 over perturbation draws; only sanity is asserted
        assert abs(la - lb) < (1e-6 * max(1.0, abs(la)), 2e-3, 5e-2)[step], (step, la, lb)
        if step == 0:
            assert float((ca - cb).abs().max() / ca.abs().max()) < 1e-5, (ca, cb)  # clip coefficient and norm
            fp = a["fp"]
            for slot in fp.order:
                o, n = fp.offsets[slot], fp.slots[slot].numel()
                ua, ub = pa[o:o + n] - a["p0"][o:o + n], pb[o:o + n] - b["p0"][o:o + n]
                e = float((ua - ub).norm() / ua.norm().clamp(min=1e-30))
                assert e < 1e-4, (slot, e)
    inside = torch.zeros_like(b["g"], dtype=torch.bool)
    for lo, hi in b["ranges"]:
        inside[lo:hi] = True
    assert bool((b["g"][inside] == 7.0).all())  # never written
    assert not bool((a["g"][inside] == 7.0).any())
    # the first step's other gradients are the same kernels on the same inputs; compare after step 1 only is not possible here (three
    # steps ran), so: a fresh single step of each mode
    gs = {}
    for mode in ("materialize", "factors"):
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(state, strict=False)
        ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to(dev)), lr=1e-3, max_grad_norm=0.5, adaln_grads=mode)
        with rng_override(**{k: v for k, v in d.items() if k != "x1"}):
            ts.step(d["x1"].to(dev))
        torch.cuda.synchronize()
        gs[mode] = ts.gflat.clone()
    # the factor form takes the projections' bias gradient from the norm partial records and d(time_emb) from one all-layer launch
    # (runtime.hip: ada_all) -- another summation order for those tensors; every other gradient is the same kernel on the same inputs
    fp = ts.fp
    for slot in fp.order:
        o, n = fp.offsets[slot], fp.slots[slot].numel()
        if bool(inside[o]):
            continue
        ga, gb = gs["materialize"][o:o + n], gs["factors"][o:o + n]
        if slot in ("SINW", "T1W", "T1B") or slot.split(".")[-1] in ("G1B", "B1B", "G2B", "B2B"):  # time MLP, projection biases
            assert float((ga - gb).norm() / ga.norm().clamp(min=1e-30)) < 1e-5, slot
        else:
            assert torch.equal(ga, gb), slot


def test_deferred_reductions_equal_per_layer_reductions():
    """vbx_model.defer_reduce (round 5): without a per-stage reader (no gradient exchange) the partial-record reductions of every layer
    -- norm gamma / beta, bias column sums, qk-norm gammas -- run in two launches after layer 0 instead of one launch per layer, and
    d(time_emb) of all layers in one launch.  Same per-tensor summation order: the gradient buffer must be BIT-identical to the one a
    per-stage reader sees (on_stage given: every layer reduces before it returns), and each stage's range must be final when its
    callback fires (checked against the end-of-backward values)."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override
    from oracle import restate

    cfg = restate.Cfg(dim=128, depth=4, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=5)
    for k in state:
        if ".to_gamma." in k or ".to_beta." in k:
            state[k] = state[k] + 0.05 * torch.randn(state[k].shape, generator=torch.Generator().manual_seed(zlib.crc32(k.encode()) % 1000))
    d = _draws(77, 3, 100, 128)
    draws = {k: v for k, v in d.items() if k != "x1"}
    vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=4, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to(dev)), lr=1e-3, max_grad_norm=0.5)
    out = {}
    for factors in (True, False):
        ts.gflat.fill_(7.0)
        with rng_override(**draws):
            ts._forward_backward(d["x1"].to(dev), None, None, on_stage=None, adaln_factors=factors)
        torch.cuda.synchronize()
        deferred = ts.gflat.clone()
        seen = {}

        def cb(i, rng):  # what a per-stage reader (the bucket reducer) would pick up
            seen[i] = (rng, ts.gflat[rng[0]:rng[1]].clone())

        ts.gflat.fill_(7.0)
        with rng_override(**draws):
            ts._forward_backward(d["x1"].to(dev), None, None, on_stage=cb, adaln_factors=factors)
        torch.cuda.synchronize()
        staged = ts.gflat.clone()
        assert torch.equal(deferred, staged), float((deferred - staged).abs().max())
        assert len(seen) == cfg.depth + 2
        for i, ((lo, hi), snap) in seen.items():
            assert torch.equal(snap, staged[lo:hi]), i  # final when the callback fired
        out[factors] = staged
    assert not torch.equal(out[True], out[False])  # (the factor form leaves the adaLN weight blocks at the sentinel)


def test_gradient_norm_from_slab_reduce_partials(monkeypatch):
    """The clip norm without a second pass over the big weight gradients (round 5): the slab reduce of every layer's to_qkv / to_out /
    FeedForward weight gradients leaves per-block sums of squares (vbx_skr_job.sq, vbx_model.sq_partials); TrainStep adds them to the
    factor terms and to a pass over the small tensors only.  Against the plain pass (VBX_SUMSQ_FOLD=0) from the same weights and draws:
    the same norm and clip coefficient to fp32 rounding, identical gradients, parameters equal to rounding after the step -- and against
    torch's own norm of the materialised gradient."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override
    from oracle import restate

    cfg = restate.Cfg(dim=128, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=21)
    for k in state:
        if ".to_gamma." in k or ".to_beta." in k:
            state[k] = state[k] + 0.05 * torch.randn(state[k].shape, generator=torch.Generator().manual_seed(zlib.crc32(k.encode()) % 1000))
    d = _draws(9, 3, 88, 128)
    draws = {k: v for k, v in d.items() if k != "x1"}
    res = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("VBX_SUMSQ_FOLD", fold)
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(state, strict=False)
        ts = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to(dev)), lr=1e-3, max_grad_norm=0.5)
        assert ts.adaln_factors_apply()
        with rng_override(**draws):
            ts.step(d["x1"].to(dev))
        torch.cuda.synchronize()
        assert bool(getattr(ts, "_sq_folded", False)) == (fold == "1")
        res[fold] = (ts.coef.clone().cpu(), ts.gflat.clone(), ts.fp.flat.clone(), ts)
    (ca, ga, pa, tsa), (cb, gb, pb, _) = res["1"], res["0"]
    assert float((ca - cb).abs().max() / cb.abs().max()) < 2e-6, (ca, cb)
    inside = torch.zeros_like(ga, dtype=torch.bool)
    for lo, hi in tsa.adaln_weight_ranges():
        inside[lo:hi] = True
    assert torch.equal(ga[~inside], gb[~inside])
    assert float((pa - pb).abs().max()) < 1e-6
    # torch's norm of the full gradient: materialise the factor-form blocks (dada_l^T . temb) and take the norm on the host in fp64
    eng = tsa._last_eng
    dada, temb = eng.adaln_factor_tensors()
    full = ga.double().cpu().clone()
    for l, (lo, hi) in enumerate(tsa.adaln_weight_ranges()):
        full[lo:hi] = (dada[l].double().t() @ temb.double()).reshape(-1).cpu()
    assert abs(float(ca[1]) - float(full.norm())) < 2e-6 * float(full.norm()), (float(ca[1]), float(full.norm()))


def test_adamw_weight_decay_matches_get_optimizer(tmp_path, golden):
    """wd > 0 (optimizer.py:10-35): AdamW with decoupled decay on the ndim >= 2 parameters only.  TrainStep(wd=...) must equal torch:
    the same gradients -> clip_grad_norm_(0.5) -> torch.optim.AdamW over get_optimizer's two parameter groups; and the trainer's
    checkpoint must carry those two groups in a torch.optim.AdamW-loadable state."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override

    g = golden("small")
    draws = dict(x0=g["x0"], times=g["times"], frac_lengths=g["frac"], rand=g["rand"])
    wd, lr = 0.3, 1e-3

    def make():
        vb = vbx.VoiceBox(dim=64, num_cond_tokens=500, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(g["state"], strict=False)
        vb = vb.to(dev)
        return vb, vbx.ConditionalFlowMatcherWrapper(voicebox=vb)

    def get_optimizer(params):  # optimizer.py:10-35 restated
        params = list(params)
        wd_p, no_wd = [p for p in params if p.ndim >= 2], [p for p in params if p.ndim < 2]
        return torch.optim.AdamW([{"params": wd_p}, {"params": no_wd, "weight_decay": 0}], lr=lr, weight_decay=wd, betas=(0.9, 0.99), eps=1e-8)

    vb_r, w_r = make()
    opt = get_optimizer(w_r.parameters())
    vb, w = make()
    ts = TrainStep(w, lr=lr, max_grad_norm=0.5, wd=wd)
    ref = dict(vb_r.named_parameters())
    for step in range(2):
        with rng_override(**draws):
            w_r(g["x1"].to(dev)).backward()
        torch.nn.utils.clip_grad_norm_([p for p in w_r.parameters() if p.grad is not None], 0.5)
        opt.step(); opt.zero_grad()
        with rng_override(**draws):
            ts.step(g["x1"].to(dev))
        # Step 1: both paths start from the same weights, run the same kernels and get the same gradients -- the updates may differ by
        # the fp32 operation order of the two Adam implementations only.  Step 2 starts from weights that differ in their last bits, and
        # at this (reference-init, near-one-hot softmax) point the gradient of a tensor below both attentions is a chaotic function of
        # them: the update of transformer.register_tokens differed by 3.0 - 4.3 % over three builds of the kernels (rounds 4 / 5) -- a
        # noise figure, so the second step is held to 10 % and the exactness claim rests on the first.
        tol = 2e-4 if step == 0 else 1e-1
        for k, p in vb.named_parameters():
            if p.requires_grad:
                upd, upd_r = p.detach() - g["state"][k].to(dev), ref[k].detach() - g["state"][k].to(dev)
                e = float((upd - upd_r).norm() / upd_r.norm().clamp(min=1e-20))
                assert e < tol, (step, k, e)
    # the decay really acted: a 2-D weight moved differently from an undecayed run
    vb0, w0 = make()
    ts0 = TrainStep(w0, lr=lr, max_grad_norm=0.5)
    with rng_override(**draws):
        ts0.step(g["x1"].to(dev))
    with rng_override(**draws):
        ts0.step(g["x1"].to(dev))
    assert float((vb0.to_pred.weight.detach() - vb.to_pred.weight.detach()).abs().max()) > 1e-5

    class Latents(torch.utils.data.Dataset):
        def __len__(self):
            return 8

        def __getitem__(self, i):
            return torch.randn(40, 64, generator=torch.Generator().manual_seed(i))

    vb2, w2 = make()
    tr = vbx.VoiceBoxTrainer(w2, batch_size=2, dataset=Latents(), num_train_steps=2, wd=wd, valid_frac=0.25, log_every=1,
                             save_results_every=100, save_model_every=1, results_folder=str(tmp_path), force_clear_prev_results=True)
    tr.train()
    pkg = torch.load(str(tmp_path / "voicebox.1.pt"), map_location="cpu")
    groups = pkg["optim"]["param_groups"]
    assert len(groups) == 2 and groups[0]["weight_decay"] == wd and groups[1]["weight_decay"] == 0
    ref_opt = get_optimizer([torch.nn.Parameter(p.detach().cpu().clone()) for p in w2.parameters()])
    ref_opt.load_state_dict(pkg["optim"])
    n_wd = sum(1 for p in w2.parameters() if p.ndim >= 2)
    assert len(groups[0]["params"]) == n_wd


def test_trainer_accumulation_and_reference_checkpoint_format(tmp_path, golden):
    """VoiceBoxTrainer (trainer.py:60-321 mirror): gradient accumulation equals one step on the concatenated batch, and the
    checkpoint is the reference's {'model','optim','scheduler'} with a torch.optim.Adam-loadable optimizer state."""
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override

    g = golden("small")

    def make():
        vb = vbx.VoiceBox(dim=64, num_cond_tokens=500, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(g["state"], strict=False)
        return vb.to(dev), None

    x1 = g["x1"].to(dev)
    draws = dict(x0=g["x0"], times=g["times"], frac_lengths=g["frac"], rand=g["rand"])
    # (a) two micro-batches of 1, weights 1/2  ==  one batch of 2 (loss = mean over the batch of per-sample masked means)
    vb_a, _ = make()
    ts_a = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb_a), lr=1e-3)
    for i in range(2):
        with rng_override(**{k: v[i:i + 1] for k, v in draws.items()}):
            ts_a.accumulate(x1[i:i + 1], 0.5)
    ts_a.apply_accumulated()
    vb_b, _ = make()
    ts_b = TrainStep(vbx.ConditionalFlowMatcherWrapper(voicebox=vb_b), lr=1e-3)
    with rng_override(**draws):
        ts_b.step(x1)
    pb = dict(vb_b.named_parameters())
    for k, p in vb_a.named_parameters():
        if p.requires_grad:
            ua, ub = p.detach() - g["state"][k].to(dev), pb[k].detach() - g["state"][k].to(dev)
            assert float((ua - ub).norm() / ub.norm().clamp(min=1e-20)) < 5e-2, k

    # (b) the driver: 3 steps with accumulation, checkpoints every step
    class Latents(torch.utils.data.Dataset):
        def __len__(self):
            return 16

        def __getitem__(self, i):
            return torch.randn(40, 64, generator=torch.Generator().manual_seed(i))

    vb, _ = make()
    wrapper = vbx.ConditionalFlowMatcherWrapper(voicebox=vb)
    tr = vbx.VoiceBoxTrainer(wrapper, batch_size=2, dataset=Latents(), num_train_steps=3, num_warmup_steps=2, grad_accum_every=2,
                             valid_frac=0.25, log_every=1, save_results_every=2, save_model_every=1, results_folder=str(tmp_path),
                             force_clear_prev_results=True)
    losses = []
    tr.train(log_fn=lambda logs: losses.append(logs["loss"]))
    assert len(losses) == 3 and all(l == l and l < 10 for l in losses)
    ck = tmp_path / "voicebox.2.pt"
    assert ck.exists()
    pkg = torch.load(str(ck), map_location="cpu")
    assert set(pkg) == {"model", "optim", "scheduler"}
    assert set(pkg["model"]) == set(wrapper.state_dict())
    # loadable by the reference's optimizer / scheduler objects
    ref_params = [torch.nn.Parameter(p.detach().cpu().clone()) for p in wrapper.parameters()]
    opt = torch.optim.Adam(ref_params, lr=3e-4, betas=(0.9, 0.99))
    opt.load_state_dict(pkg["optim"])
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=3)
    sched.load_state_dict(pkg["scheduler"])
    n_state = sum(1 for p in wrapper.parameters() if p.requires_grad)
    assert len(pkg["optim"]["state"]) == n_state
    # resume: Adam moments and step counter come back
    vb2, _ = make()
    tr2 = vbx.VoiceBoxTrainer(vbx.ConditionalFlowMatcherWrapper(voicebox=vb2), batch_size=2, dataset=Latents(), num_train_steps=6,
                              grad_accum_every=1, valid_frac=0.25, results_folder=str(tmp_path / "r2"))
    tr2.load(str(ck))
    assert int(tr2.steps.item()) == 3 and tr2.train_step_fn.steps == 3
    assert torch.equal(tr2.train_step_fn.m, tr.train_step_fn.m) and torch.equal(tr2.train_step_fn.v, tr.train_step_fn.v)
    for (k, a), (_, b) in zip(vb2.state_dict().items(), vb.state_dict().items()):
        assert torch.equal(a, b), k
    tr2.train_step()
    # split_batches=True (trainer.py:83,93): batch_size is the global batch; one process -> the same loader
    vb3, _ = make()
    tr3 = vbx.VoiceBoxTrainer(vbx.ConditionalFlowMatcherWrapper(voicebox=vb3), batch_size=2, dataset=Latents(), num_train_steps=2,
                              valid_frac=0.25, results_folder=str(tmp_path / "r3"), split_batches=True)
    assert tr3.rank_batch_size == 2 and tr3.dl.batch_size == 2
    tr3.train_step()


def _nccl_world1_worker(port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["VBX_FORCE_DIST"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))  # nccl == RCCL on ROCm
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import TrainStep
    from voicebox_pytorch_amd.masks import rng_override
    from oracle import restate

    cfg = restate.Cfg(dim=128, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=5)
    res = {}
    for forced in ("1", "0"):
        os.environ["VBX_FORCE_DIST"] = forced
        vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
        vb.load_state_dict(state, strict=False)
        wrapper = vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to("cuda:0"))
        ts = TrainStep(wrapper, lr=1e-3, max_grad_norm=0.5, bucket_bytes=1 << 16)
        d = _draws(100, 2, 72, 128)
        launched = 0
        flat1 = None
        for _ in range(3):
            with rng_override(**{k: v for k, v in d.items() if k != "x1"}):
                loss = ts.step(d["x1"].cuda())
            if flat1 is None:
                torch.cuda.synchronize()
                flat1, g1 = ts.fp.flat.detach().cpu().clone(), ts.gflat.detach().cpu().clone()
        torch.cuda.synchronize()
        outside = torch.ones(ts.gflat.numel(), dtype=torch.bool)
        for lo, hi in ts.adaln_weight_ranges():
            outside[lo:hi] = False
        res[forced] = dict(loss=float(loss), flat=ts.fp.flat.detach().cpu().clone(), g=ts.gflat.detach().cpu().clone(),
                           exchange=bool(ts.exchange), comm_stream=ts.comm_stream is not None, outside=outside, flat1=flat1, g1=g1,
                           wire_bytes=getattr(ts, "wire_bytes", None))
    # shard mode over RCCL: the in-place reduce_scatter_tensor / all_gather_into_tensor path must be the one that runs (ADVICE r4:
    # the gloo tests only ever see the all-reduce fallback)
    os.environ["VBX_FORCE_DIST"] = "1"
    vb = vbx.VoiceBox(dim=128, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    wrapper = vbx.ConditionalFlowMatcherWrapper(voicebox=vb.to("cuda:0"))
    ts = TrainStep(wrapper, lr=1e-3, max_grad_norm=0.5, bucket_bytes=1 << 16, grad_mode="shard")
    d = _draws(100, 2, 72, 128)
    flat1 = None
    for _ in range(3):
        with rng_override(**{k: v for k, v in d.items() if k != "x1"}):
            loss = ts.step(d["x1"].cuda())
        if flat1 is None:
            torch.cuda.synchronize()
            flat1 = ts.fp.flat.detach().cpu().clone()
    torch.cuda.synchronize()
    res["shard"] = dict(loss=float(loss), flat=ts.fp.flat.detach().cpu().clone(), flat1=flat1, used_reduce_scatter=bool(ts._red.used_reduce_scatter),
                        native=bool(ts._red.native_shard_collectives))
    dist.destroy_process_group()
    torch.save(res, out)


def test_rccl_path_executes_at_world_size_1(tmp_path):
    """The RCCL code path of dp.TrainStep (init_process_group('nccl'), bucketed async all-reduce on the comm stream, work.wait(),
    clip + Adam) on the ONE GPU of the test box: VBX_FORCE_DIST=1 issues the collectives at world size 1, where an all-reduce is
    the identity -- three steps must give bit-identical parameters to the same steps without any collective."""
    out = str(tmp_path / "nccl1.pt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), out))
    p.start()
    p.join(300)
    assert p.exitcode == 0, p.exitcode
    res = torch.load(out)
    assert res["1"]["exchange"] and res["1"]["comm_stream"] and not res["0"]["exchange"]
    assert abs(res["1"]["loss"] - res["0"]["loss"]) < 2e-3  # third step of two runs whose adaLN weights differ in their last bits
    assert res["shard"]["native"] and res["shard"]["used_reduce_scatter"], res["shard"]
    # the FIRST step of every variant starts from the same weights: same update up to the last bits of the clip coefficient and the
    # operation order of the adaLN blocks' gradient (expanded into the buffer vs inside Adam); later steps are chaotic at this init
    # (the shard run materialises the adaLN weight gradients: its projection-bias and time-MLP gradients are summed in another order
    #  -- runtime.hip: ada_all -- and the first Adam step turns a last-bit change of a tiny gradient into up to ~2e-6 of update)
    assert float((res["shard"]["flat1"] - res["0"]["flat1"]).abs().max()) < 5e-6
    assert float((res["1"]["flat1"] - res["0"]["flat1"]).abs().max()) < 2e-6
    # forced = "1": the exchange is active, the adaLN weight gradients travel as factors and are expanded into the buffer; forced = "0":
    # one GPU without exchange, they stay in factor form and their blocks of the buffer are not written -- compare everything else
    # bit for bit, and the parameters (whose adaLN blocks were updated from the same factors by two different kernels) to rounding
    outside = res["0"]["outside"]
    assert torch.equal(res["1"]["g1"][outside], res["0"]["g1"][outside])  # first step: the same kernels on the same weights
    # (no assertion on the parameters after the THIRD step: the two runs' adaLN weights differ in their last bits after step 1 and at this
    #  init single elements of a later Adam update flip sign -- measured max difference 1.5e-3 = one step of lr; the first-step check below
    #  is the exact one)
