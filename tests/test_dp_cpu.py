"""CPU (-m "not gpu"): the data-parallel gradient exchange (voicebox_pytorch_amd.dp.GradBucketReducer) under
gloo with world_size 2: bucketed all-reduce of the flat gradient buffer, fed stage by stage in backward order,
equals the single-process gradient on the concatenated batch (DDP semantics, trainer.py:89-95,270)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_flat_grads(fp, cfg, state, x1, x0, times, frac, rand):
    from oracle import restate

    # fp64: the qk-normed attention gradient is so ill-conditioned that fp32 rounding differences between a batch-2
    # and a batch-4 matmul already move it by ~1e-3 relative; in fp64 the DDP identity holds to round-off
    p = {k: v.double().clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in state.items()}
    loss = restate.cfm_loss(p, cfg, x1.double(), x0.double(), times.double(), frac, rand)
    loss.backward()
    g = torch.zeros(fp.numel)
    name_of = {id(prm): name for name, prm in fp._named}
    for slot in fp.order:
        prm = fp.slots[slot]
        o = fp.offsets[slot]
        g[o:o + prm.numel()] = p[name_of[id(prm)]].grad.flatten().float()
    return g


def _worker(rank, world, port, bucket_bytes, out, comm_dtype=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import GradBucketReducer
    from oracle import restate

    cfg = restate.Cfg(dim=64, depth=2, heads=2, dim_head=64)
    state = restate.init_state_dict(cfg, seed=7)
    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    vb.load_state_dict(state, strict=False)
    fp = vb.flat_params()
    fp._named = list(vb.named_parameters())
    g = torch.Generator().manual_seed(3)
    B, N = 4, 24
    x1, x0 = torch.randn(B, N, 64, generator=g), torch.randn(B, N, 64, generator=g)
    times, frac, rand = torch.rand(B, generator=g), 0.7 + 0.3 * torch.rand(B, generator=g), torch.rand(B, generator=g)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    gflat = _oracle_flat_grads(fp, cfg, state, x1[sl], x0[sl], times[sl], frac[sl], rand[sl])
    red = GradBucketReducer(gflat, fp.stage_ranges, bucket_bytes=bucket_bytes, comm_dtype=comm_dtype)
    for i, rng in enumerate(fp.stage_ranges):  # backward order: head, layer L-1 .. 0, embed
        red.stage_done(i, rng)
    red.finish()
    gflat /= world
    if rank == 0:
        full = _oracle_flat_grads(fp, cfg, state, x1, x0, times, frac, rand)
        out.put((float((gflat - full).abs().max()), float(full.abs().max()), len(red.buckets_launched),
                 red.buckets_launched[0][0], red.buckets_launched[-1][1], fp.numel))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [1, 1 << 30])
def test_bucketed_allreduce_equals_full_batch_gradient(bucket_bytes):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    err, scale, nbuckets, lo, hi, numel = res
    assert err < 1e-5 * max(scale, 1.0), res
    assert lo == 0 and hi == numel  # buckets cover the whole flat buffer
    assert nbuckets == (4 if bucket_bytes == 1 else 1)  # depth 2: head, 2 layers, embed -> 4 stages


def test_bf16_gradient_exchange_option():
    """comm_dtype=torch.bfloat16 (optional wire compression): the reduced gradient equals the exact one to bf16 precision."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 1 << 30, out, torch.bfloat16)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    err, scale, nbuckets, lo, hi, numel = res
    assert err < 2 ** -7 * max(scale, 1e-6), res  # two bf16 roundings of values <= scale
    assert lo == 0 and hi == numel


def test_warmup_cosine_schedule_matches_trainer_rule():
    """WarmupCosineLR replays VoiceBoxTrainer's rule (trainer.py:231-253): manual linear warm-up written into the optimizer,
    then torch's CosineAnnealingLR.step() once per step."""
    from torch.optim.lr_scheduler import CosineAnnealingLR

    from voicebox_pytorch_amd.dp import WarmupCosineLR

    for lr, init, warm, total in ((3e-4, 1e-5, 7, 40), (1e-4, 1e-5, 0, 25), (5e-4, 0.0, 3, 10)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=lr)
        sched = CosineAnnealingLR(opt, T_max=total)
        mine = WarmupCosineLR(lr, total, warm, init)
        for step in range(total + 5):
            if step < warm:  # trainer.py:241-247
                for g in opt.param_groups:
                    g["lr"] = init + (lr - init) * step / warm
            else:
                opt.step()
                sched.step()
            want = opt.param_groups[0]["lr"]
            got = mine.rate_for_step(step)
            assert abs(got - want) <= 1e-12 + 1e-9 * abs(want), (step, got, want)


def test_last_bucket_is_small_and_staging_is_persistent():
    """the last bucket of the gradient exchange cannot overlap any backward work, so it must hold the embed /
    time-MLP stage only, not the last two layers as well; and the wire-dtype staging buffer is allocated once, not per bucket per step.
    Bucket policy replayed on the host (world size 1: the collectives are skipped, the bucket boundaries are what is checked)."""
    from voicebox_pytorch_amd.dp import GradBucketReducer

    layer, head, embed = 8_500_000, 300_000, 1_200_000  # floats: dim 512 / depth 12 proportions (34 MB layers, 4.8 MB embed stage)
    bounds = [0, head] + [head + layer * (i + 1) for i in range(12)]
    bounds.append(bounds[-1] + embed)
    ranges = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    g = torch.zeros(bounds[-1])
    red = GradBucketReducer(g, ranges, bucket_bytes=130 << 20, tail_bytes=16 << 20)  # 130 MiB: five 34 MB layers per bucket, two left over
    for i, rng in enumerate(ranges):
        red.stage_done(i, rng)
    red.finish()
    b = red.buckets_launched
    assert b[0][0] == 0 and b[-1][1] == bounds[-1] and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))  # a partition, in order
    assert b[-1] == ranges[-1], b[-3:]                               # the un-overlapped bucket is the embed stage alone
    assert (b[-1][1] - b[-1][0]) * 4 <= 16 << 20
    assert all((hi - lo) * 4 >= 130 << 20 for lo, hi in b[:-2])       # the others still fill up (the one before the tail may be short)
    # tail policy off -> the old behaviour: the final bucket swallows the last layers
    red0 = GradBucketReducer(g, ranges, bucket_bytes=130 << 20, tail_bytes=0)
    for i, rng in enumerate(ranges):
        red0.stage_done(i, rng)
    red0.finish()
    assert (red0.buckets_launched[-1][1] - red0.buckets_launched[-1][0]) * 4 > 16 << 20
    # staging: one persistent buffer, reused across buckets and across reducers
    red1 = GradBucketReducer(g, ranges, bucket_bytes=64 << 20, comm_dtype=torch.bfloat16)
    v1 = red1._stage(*ranges[0])
    v2 = red1._stage(*ranges[1])
    assert v1.dtype == torch.bfloat16 and v1.untyped_storage().data_ptr() == v2.untyped_storage().data_ptr()
    red2 = GradBucketReducer(g, ranges, bucket_bytes=64 << 20, comm_dtype=torch.bfloat16, stage_buf=red1.stage_buf)
    assert red2._stage(*ranges[2]).untyped_storage().data_ptr() == v1.untyped_storage().data_ptr()


def _shard_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from voicebox_pytorch_amd.dp import GradBucketReducer

    # a flat buffer in this package's layout (64-float slots), three stages; rank-specific gradients
    bounds = [0, 64 * 5, 64 * 5 + 64 * 40, 64 * 5 + 64 * 40 + 64 * 7]
    ranges = [(bounds[i], bounds[i + 1]) for i in range(3)]
    n = bounds[-1]
    gen = torch.Generator().manual_seed(100 + rank)
    g_local = torch.randn(n, generator=gen)
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(7))
    lr, b1, b2, eps, max_norm = 1e-3, 0.9, 0.99, 1e-8, 0.5

    def adam(p, g, m, v, coef):  # torch.optim.Adam, step 1, gradient pre-multiplied by the clip coefficient
        gr = g * coef
        m.mul_(b1).add_(gr, alpha=1 - b1)
        v.mul_(b2).addcmul_(gr, gr, value=1 - b2)
        p.sub_(lr / (1 - b1) * m / (v.sqrt() / (1 - b2) ** 0.5 + eps))

    def clip_coef(sumsq):
        norm = sumsq.sqrt() / world
        return torch.clamp(max_norm / (norm + 1e-6), max=1.0) / world

    # A: replicated (all-reduce, every rank updates everything)
    gA = g_local.clone()
    redA = GradBucketReducer(gA, ranges, bucket_bytes=64 * 30 * 4)
    for i, rng in enumerate(ranges):
        redA.stage_done(i, rng)
    redA.finish()
    # B: sharded (reduce-scatter, chunk owners update, all-gather)
    gB = g_local.clone()
    redB = GradBucketReducer(gB, ranges, bucket_bytes=64 * 30 * 4, shard=True)
    for i, rng in enumerate(ranges):
        redB.stage_done(i, rng)
    redB.finish()
    assert redB.buckets_launched == redA.buckets_launched and len(redB.owned) == len(redB.buckets_launched)
    own_ok = all(torch.equal(gB[lo:hi], gA[lo:hi]) for lo, hi in redB.owned)  # the owned chunk holds the all-reduce's sum, bit for bit
    part = torch.stack([gB[lo:hi].pow(2).sum() for lo, hi in redB.owned]).sum().reshape(1)
    allp = torch.zeros(world)
    dist.all_gather_into_tensor(allp, part)
    coefB = clip_coef(allp.sum())
    coefA = clip_coef(gA.pow(2).sum())
    pA, mA, vA = p0.clone(), torch.zeros(n), torch.zeros(n)
    adam(pA, gA, mA, vA, coefB)  # the same coefficient: the update itself must then be bit-identical
    pB, mB, vB = p0.clone(), torch.zeros(n), torch.zeros(n)
    for lo, hi in redB.owned:
        adam(pB[lo:hi], gB[lo:hi], mB[lo:hi], vB[lo:hi], coefB)
    redB.all_gather(pB)
    redB.all_gather(mB)
    redB.all_gather(vB)
    covered = sum(hi - lo for lo, hi in redB.owned)
    if rank == 0:
        out.put((own_ok, torch.equal(pA, pB), torch.equal(mA, mB) and torch.equal(vA, vB), float(abs(coefA - coefB) / coefA), covered, n))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_exchange_equals_replicated_bit_for_bit():
    """grad_mode="shard" (reduce-scatter -> owners clip / Adam their 1 / world -> all-gather) against the replicated exchange, gloo
    world 2, fp32: the owned chunks hold the all-reduce's sums bit for bit, the gathered parameters and moments equal the replicated
    update bit for bit given the clip coefficient, and the coefficient itself (a sum of chunk partials in rank order instead of one
    sum) agrees to fp32 rounding."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    own_ok, p_same, mv_same, dcoef, covered, n = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert own_ok and p_same and mv_same
    assert dcoef < 1e-6, dcoef
    assert covered * 2 == n  # each rank owns exactly half of the flat buffer


def _factor_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from voicebox_pytorch_amd.dp import GradBucketReducer, gather_adaln_factors

    # a flat buffer in this package's layout: head | layer 1 = [adaLN weight block | rest] | layer 0 = [adaLN weight block | rest] | embed
    L, B, J4, Th = 2, 3, 128, 192
    blk = J4 * Th
    sizes = [64 * 5, blk + 64 * 9, blk + 64 * 7, 64 * 4]
    bounds = [0]
    for z in sizes:
        bounds.append(bounds[-1] + z)
    ranges = [(bounds[i], bounds[i + 1]) for i in range(4)]
    ada = {1: (bounds[1], bounds[1] + blk), 0: (bounds[2], bounds[2] + blk)}  # layer -> its adaLN weight block
    n = bounds[-1]
    gen = torch.Generator().manual_seed(100 + rank)
    dada, temb = torch.randn(L, B, J4, generator=gen), torch.randn(B, Th, generator=gen)
    g_local = torch.randn(n, generator=gen)
    for l, (lo, hi) in ada.items():
        g_local[lo:hi] = (dada[l].t() @ temb).flatten()  # what the backward writes in "materialize" mode
    # A: every gradient all-reduced (DDP)
    gA = g_local.clone()
    redA = GradBucketReducer(gA, ranges, bucket_bytes=64 * 30 * 4)
    for i, rng in enumerate(ranges):
        redA.stage_done(i, rng)
    redA.finish()
    # B: the adaLN weight blocks travel as factors; their slots of the local buffer were never written (NaN here)
    gB = g_local.clone()
    for lo, hi in ada.values():
        gB[lo:hi] = float("nan")
    redB = GradBucketReducer(gB, ranges, bucket_bytes=64 * 30 * 4, skip_ranges=list(ada.values()))
    for i, rng in enumerate(ranges):
        redB.stage_done(i, rng)
    dada_all, temb_all, fwire = gather_adaln_factors(dada, temb)
    redB.finish()
    for l, (lo, hi) in ada.items():
        gB[lo:hi] = (dada_all[l].t() @ temb_all).flatten()  # vbx_adaln_expand_dw on the GPU
    covered = sorted(redB.buckets_launched)
    gaps_ok = all(not (lo < b and a < hi) for lo, hi in covered for a, b in ada.values())  # no bucket touches a factor block
    lr, b1, b2, eps = 1e-3, 0.9, 0.99, 1e-8

    def adam(g):
        p = torch.randn(n, generator=torch.Generator().manual_seed(7))
        coef = torch.clamp(0.5 / (g.pow(2).sum().sqrt() / world + 1e-6), max=1.0) / world
        gr = g * coef
        m, v = (1 - b1) * gr, (1 - b2) * gr * gr
        return p - lr / (1 - b1) * m / (v.sqrt() / (1 - b2) ** 0.5 + eps)

    pA, pB = adam(gA), adam(gB)
    if rank == 0:
        out.put((float((gA - gB).abs().max() / gA.abs().max()), float((pA - pB).abs().max()), gaps_ok, redA.wire_floats,
                 redB.wire_floats + fwire, sum(hi - lo for lo, hi in covered), n - 2 * blk, bool(torch.isfinite(gB).all())))
    dist.barrier()
    dist.destroy_process_group()


def test_adaln_factor_exchange_equals_allreduce_of_the_products():
    """TrainStep(adaln_grads="factors") at world size > 1: the adaLN weight blocks are left out of the bucketed
    all-reduce (GradBucketReducer(skip_ranges=...)), every rank all-gathers the factors dada [L, B, 4 D] / temb [B, Th] and expands
    dada_all^T . temb_all locally.  gloo, world 2: the resulting gradient equals the all-reduce of the materialised products to fp32
    rounding, so does the parameter after one clipped Adam step; no bucket touches a factor block; the wire carries the rest of the
    buffer plus B * (L * 4 D + Th) floats per rank instead of the whole buffer."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_factor_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    gerr, perr, gaps_ok, wireA, wireB, covered, rest, finite = out.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert finite and gaps_ok and covered == rest
    assert gerr < 1e-6 and perr < 1e-6, (gerr, perr)
    assert wireB < 0.1 * wireA, (wireA, wireB)  # here the factor blocks are 97 % of the buffer; at dim 512 / depth 12 they are 49 %


def _wide_worker(rank, world, port, out):
    """One rank of the world-8 / world-3 exchange test: the REAL flat layout of the dim-64 model (stage ranges, adaLN weight blocks),
    rank-seeded gradients and factors; replicated + factor exchange, then shard mode."""
    import contextlib
    import io

    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import voicebox_pytorch_amd as vbx
    from voicebox_pytorch_amd.dp import GradBucketReducer, gather_adaln_factors

    vb = vbx.VoiceBox(dim=64, num_cond_tokens=5, depth=2, dim_head=64, heads=2, condition_on_text=False)
    fp = vb.flat_params()
    n, ranges = fp.numel, fp.stage_ranges
    ada = sorted((fp.offsets[f"L{l}.G1W"], fp.offsets[f"L{l}.B2W"] + fp.slots[f"L{l}.B2W"].numel()) for l in range(fp.depth))
    L, B, J4 = fp.depth, 3, 4 * 64
    Th = (ada[0][1] - ada[0][0]) // J4
    assert J4 * Th == ada[0][1] - ada[0][0]

    def local(r):  # rank r's gradient buffer and factors (any rank can rebuild any other's: the expected sums need no collective)
        gen = torch.Generator().manual_seed(100 + r)
        return torch.randn(n, generator=gen), torch.randn(L, B, J4, generator=gen), torch.randn(B, Th, generator=gen)

    g_local, dada, temb = local(rank)
    everyone = [local(r) for r in range(world)]
    want = torch.stack([e[0] for e in everyone]).sum(0)
    want_blocks = {l: sum(e[1][l].t() @ e[2] for e in everyone).flatten() for l in range(L)}
    bucket = 4 * (n // 5)  # several buckets, boundaries at stage ends
    # A: replicated exchange, the adaLN weight blocks as factors
    gA = g_local.clone()
    for lo, hi in ada:
        gA[lo:hi] = float("nan")  # never written in factor mode
    redA = GradBucketReducer(gA, ranges, bucket_bytes=bucket, skip_ranges=ada)
    for i, rng in enumerate(ranges):
        redA.stage_done(i, rng)
    dada_all, temb_all, fwire = gather_adaln_factors(dada, temb)
    redA.finish()
    order = sorted(range(L), key=lambda l: fp.offsets[f"L{l}.G1W"])  # ada[i] belongs to layer order[i]
    for (lo, hi), l in zip(ada, order):
        gA[lo:hi] = (dada_all[l].t() @ temb_all).flatten()
    errA = 0.0
    inside = torch.zeros(n, dtype=torch.bool)
    for (lo, hi), l in zip(ada, order):
        inside[lo:hi] = True
        errA = max(errA, float((gA[lo:hi] - want_blocks[l]).abs().max() / want_blocks[l].abs().max()))
    errA = max(errA, float((gA[~inside] - want[~inside]).abs().max() / want.abs().max()))
    touches = any(lo < b and a < hi for lo, hi in redA.buckets_launched for a, b in ada)
    # B: shard mode on the whole buffer
    gB = g_local.clone()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        redB = GradBucketReducer(gB, ranges, bucket_bytes=bucket, shard=True)
        for i, rng in enumerate(ranges):
            redB.stage_done(i, rng)
        redB.finish()
    own_err = max(float((gB[lo:hi] - want[lo:hi]).abs().max()) for lo, hi in redB.owned if hi > lo)
    p = torch.full((n,), float("nan"))
    for lo, hi in redB.owned:
        p[lo:hi] = want[lo:hi] * 0.5  # "the owners' update"
    redB.all_gather(p)
    gathered_ok = bool(torch.allclose(p, want * 0.5, rtol=1e-6, atol=1e-6))
    sizes = torch.tensor([float(sum(hi - lo for lo, hi in redB.owned))])
    alls = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(alls, sizes)
    even = all((hi - lo) % world == 0 for lo, hi in redB.buckets_launched)
    if rank == 0:
        out.put((errA, touches, len(redA.buckets_launched), own_err / float(want.abs().max()), gathered_ok,
                 int(sum(float(a) for a in alls)), n, even, redB.uneven_logged, "does not divide by world size" in buf.getvalue(),
                 redA.wire_floats + fwire, n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [8, 3])
def test_exchange_at_world_8_and_uneven_world_3(world):
    """the exchange arithmetic that the first 8-GPU run will execute, on gloo with the dim-64 model's real flat
    layout.  World 8: bucket boundaries and chunk_of at W = 8 (every bucket divides evenly: flat slots are multiples of 64 floats),
    the bucketed all-reduce with the adaLN weight blocks skipped + the factor all-gather reproduces sum_r g_r outside the blocks and
    sum_r dada_r^T . temb_r inside, shard mode's owned chunks hold the all-reduce's sums, partition the buffer exactly once over the
    ranks and the parameter all-gather restores every chunk.  World 3: buckets do not divide by 3 -- shard mode takes the all-reduce +
    per-chunk-broadcast fallback, says so once on rank 0, and still delivers the same sums / partition / gather."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    errA, touches, nb, own_err, gathered_ok, covered, n, even, uneven_logged, said_so, wire, total = out.get(timeout=600)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert errA < 1e-5, errA          # fp32 sums of `world` terms in another order
    assert not touches and nb >= 3    # no bucket touches a factor block; the stage order produced several buckets
    assert own_err < 1e-5 and gathered_ok
    assert covered == n               # the owned chunks of all ranks partition the buffer
    if world == 8:
        assert even and not uneven_logged
    else:
        assert not even and uneven_logged and said_so
    assert wire < 0.75 * total        # the factor blocks never travel as products
