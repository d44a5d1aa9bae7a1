"""Headline benchmark: mel-frames/sec of a full CFM train step (forward + backward + gradient all-reduce +
clip + Adam) of VoiceBox dim 512 / depth 12 / heads 16 on synthetic (B=8 per GPU, 1024 frames, dim 512)
-- BASELINE.json `metric`, configs[3] (the largest single-GPU configuration at N=1; weak scaling for N>1).

    python bench.py --gpus N --steps K --warmup W          (N>1 without a launcher: re-executes itself as N ranks through
                                                            torch.distributed.run; under a launcher it reads RANK / WORLD_SIZE)

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  roofline     : the dominant kernel's achieved TFLOP/s (algorithmic FLOPs / HIP-event-measured launch time)
                 against the 2.5 PFLOP/s dense bf16/fp16 MFMA peak, plus the whole-step fraction;
  cpu_baseline : the CPU oracle (oracle/restate.py, kind "port") timed on this box's host cores on a bounded sample.
`--mode sample` benchmarks ConditionalFlowMatcherWrapper.sample (64 midpoint intervals = 128 NFE) instead.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: needed by RCCL on this driver for multi-process runs

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


def fwd_flops_per_frame(D, L, H, N, R, ff_mult=4):
    """Algorithmic forward FLOPs per mel frame (SURVEY 8(a)/(d) formulas, 2*MAC, per batch element)."""
    I, F, Np, Th = H * 64, int(D * ff_mult * 2 / 3), N + R, 4 * D
    per_layer = 2 * Np * D * 3 * I + 2 * 2 * H * Np * Np * 64 + 2 * Np * I * D + 2 * Np * D * 2 * F + 2 * Np * F * D + 4 * 2 * Th * D
    other = 2 * N * 2 * D * D + 2 * N * D * D + 2 * N * D * 31 + 2 * D * Th
    return (L * per_layer + other) / N


def build_model(args, dev):
    import voicebox_pytorch_amd as vbx

    torch.manual_seed(0)  # identical weights on every rank
    vb = vbx.VoiceBox(dim=args.dim, num_cond_tokens=500, depth=args.depth, dim_head=64, heads=args.heads,
                      condition_on_text=False, use_gateloop_layers=args.gateloop, attn_dropout=args.attn_dropout,
                      ff_dropout=args.ff_dropout)
    with torch.no_grad():  # exercise the time conditioning (adaLN projections are zero-initialised)
        for name, p in vb.named_parameters():
            if ".to_gamma.weight" in name or ".to_beta." in name:
                p.normal_(0.0, 0.02)
    vb = vb.to(dev)
    return vbx, vb, vbx.ConditionalFlowMatcherWrapper(voicebox=vb)


def time_kernel(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3  # seconds per launch


def stage_flops(args):
    """Algorithmic FLOPs per launch of every MFMA stage the runtime labels (SURVEY 8(a) formulas at the benchmark shape;
    attention backward = 2x forward: the S / dP recompute is NOT counted as useful work)."""
    B, N, D, H, R = args.batch, args.frames, args.dim, args.heads, 16
    Np, I, F = N + R, H * 64, int(D * 4 * 2 / 3)
    M = B * Np
    qkv, out, ffi, ffo = 2.0 * M * D * 3 * I, 2.0 * M * I * D, 2.0 * M * D * 2 * F, 2.0 * M * F * D
    att = 2.0 * 2 * B * H * Np * Np * 64
    return {"fwd to_qkv": qkv, "fwd attention": att, "fwd to_out": out, "fwd ff_in": ffi, "fwd ff_out": ffo,
            "dgrad ff_out": ffo, "dgrad ff_in": ffi, "dgrad to_out": out, "dgrad to_qkv": qkv, "bwd attention": 2 * att,
            "wgrad (4 GEMMs)": qkv + out + ffi + ffo}


class ProfEntry(__import__("ctypes").Structure):
    import ctypes as _C
    _fields_ = [("label", _C.c_char * 24), ("calls", _C.c_int), ("total_us", _C.c_float)]


def in_situ_stage_table(args, step, n_steps):
    """Per-stage launch times measured IN SITU: the runtime brackets every MFMA stage with HIP events on the launch stream
    (vbx_prof_enable / vbx_prof_collect, include/vbx.h) during `n_steps` extra steps of the same workload."""
    import ctypes as C

    from voicebox_pytorch_amd import _lib

    lib = _lib.lib()
    lib.vbx_prof_collect.argtypes = [C.POINTER(ProfEntry), C.c_int]
    lib.vbx_prof_collect.restype = C.c_int
    torch.cuda.synchronize()
    lib.vbx_prof_enable(1)
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    tab = (ProfEntry * 32)()
    n = lib.vbx_prof_collect(tab, 32)
    fl = stage_flops(args)
    rows = []
    for e in tab[:max(n, 0)]:
        label = e.label.decode()
        us = e.total_us / max(e.calls, 1)
        row = {"stage": label, "us_per_launch": round(us, 2), "launches_per_step": round(e.calls / n_steps, 2),
               "us_per_step": round(e.total_us / n_steps, 1)}
        if label in fl:
            row["gflop_per_launch"] = round(fl[label] / 1e9, 2)
            row["tflops"] = round(fl[label] / us / 1e6, 1)
            row["frac"] = round(fl[label] / us / 1e6 / PEAK_MFMA_TFLOPS, 4)
        rows.append(row)
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows


def isolated_gemm_us(args, dev, which):
    """HIP-event time of ONE GEMM stage launched back to back through the C ABI exactly as the runtime launches it."""
    from voicebox_pytorch_amd import _lib

    B, N, D, R, H = args.batch, args.frames, args.dim, 16, args.heads
    M, F, I = B * (N + R), int(D * 4 * 2 / 3), H * 64
    Fp = (F + 63) // 64 * 64
    lib, st = _lib.lib(), _lib.current_stream()
    d = _lib.GemmDesc()
    keep = []
    if which == "fwd ff_in":
        x = torch.randn(M, D, device=dev).half()
        w = (torch.randn(2 * Fp, D, device=dev) * D ** -0.5).half()
        bias = torch.zeros(2 * Fp, device=dev)
        g = torch.empty(M, Fp, dtype=torch.float16, device=dev)
        d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = _lib.VBX_GEMM_NT, _lib.VBX_EPI_GEGLU, M, 2 * Fp, D, D, D, Fp
        d.A, d.B, d.C, d.bias, d.f16 = x.data_ptr(), w.data_ptr(), g.data_ptr(), bias.data_ptr(), 1
        keep = [x, w, bias, g]
        if args.mode == "train":  # the SAME work as the in-situ stage: the training forward also writes the bf16 pre-activation and the bf16 copy
            h1 = torch.empty(M, 2 * Fp, dtype=torch.bfloat16, device=dev)
            gb = torch.empty(M, Fp, dtype=torch.bfloat16, device=dev)
            d.C2, d.C3 = h1.data_ptr(), gb.data_ptr()
            keep += [h1, gb]
    elif which == "dgrad to_qkv":
        x = torch.randn(M, 3 * I, device=dev).bfloat16()
        w = (torch.randn(3 * I, D, device=dev) * D ** -0.5).bfloat16()
        c = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = _lib.VBX_GEMM_NN, _lib.VBX_EPI_BF16, M, D, 3 * I, 3 * I, D, D
        d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), c.data_ptr()
        keep = [x, w, c]
    else:
        return None

    def launch():
        rc = lib.vbx_gemm(d, st)
        assert rc == 0

    return round(time_kernel(launch) * 1e6, 2)


def pmc_traffic(stage):
    """HBM bytes per launch of the stage's kernel from the COMMITTED PMC passes of this command (tools/pmc_summary.py) -- a
    constant read from profiles/, not measured in this run: returns (bytes, source file)."""
    for name in ("r06_train_pmc.json", "r05_train_pmc.json", "r04_train_pmc.json", "r03_train_pmc.json", "r02_train_pmc.json", "r02_mid_train_pmc.json", "r01_bench_pmc_hbm.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pmc):
            continue
        tab = json.load(open(pmc))
        key = {"fwd ff_in": "GEGLU", "fwd to_qkv": "QKV", "wgrad (4 GEMMs)": "gemm3_grouped", "bwd attention": "attn_bwd",
               "fwd attention": "attn_fwd"}.get(stage)
        if key is None:
            return None, None
        tot = 0.0
        for kname, e in tab.items():
            if key in kname and "hbm_bytes_per_launch" in e:
                tot += e["hbm_bytes_per_launch"]
        if tot:
            return round(tot), f"profiles/{name} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; not re-measured in this run)"
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's path, oracle/restate.py, fp32 torch CPU) on this box's host cores, on a bounded
    sample of the SAME workload: the benchmark's own per-GPU batch at 32 threads, one warm-up + three timed fwd+bwd CFM steps
    (BASELINE.md section 3).  `value` = frames/s of the best step, `median_value` of the median, `cores` = the threads used."""
    from oracle import restate

    cfg = restate.Cfg(dim=args.dim, depth=args.depth, heads=args.heads, dim_head=64)
    state = restate.init_state_dict(cfg, seed=0)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in state.items()}

    def timed_steps(Bs, n_timed):
        g = torch.Generator().manual_seed(0)
        x1, x0 = torch.randn(Bs, args.frames, args.dim, generator=g), torch.randn(Bs, args.frames, args.dim, generator=g)
        times, frac, rand = torch.rand(Bs, generator=g), 0.7 + 0.3 * torch.rand(Bs, generator=g), torch.rand(Bs, generator=g)
        ts = []
        for it in range(1 + n_timed):
            t0 = time.perf_counter()
            loss = restate.cfm_loss(p, cfg, x1, x0, times, frac, rand)
            loss.backward()
            dt = time.perf_counter() - t0
            if it > 0:
                ts.append(dt)
            for v in p.values():
                v.grad = None
        return sorted(ts)

    host = os.cpu_count() or 1
    cores = min(32, host)  # fixed: torch's CPU kernels stop scaling there on the EPYC boxes (16 / 32 / 64 threads measured in rounds 2-3:
    torch.set_num_threads(cores)  # 1065 / 1129 / 585 frames/s at batch 2); a per-run sweep made "best" flip between runs
    Bs = args.batch
    ts = timed_steps(Bs, 3)
    best, med = ts[0], ts[len(ts) // 2]
    return {"value": round(Bs * args.frames / best, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "median_value": round(Bs * args.frames / med, 1), "cpu": cpu_model(), "host_cores_total": host,
            "verified_vs_reference": "the port reproduces the unmodified reference's loss to 1e-5 and its gradients to rtol 2e-3 "
                                     "(tests/test_oracle.py::test_restatement_vs_live_reference, tests/golden/*.pt)",
            "sample": f"oracle fwd+bwd (fp32, torch CPU), batch {Bs} (the benchmark's per-GPU batch) x {args.frames} frames, dim {args.dim}, "
                      f"depth {args.depth}, {cores} threads; 1 warm-up + 3 timed steps; best {best:.2f} s/step, median {med:.2f} s/step"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec as N ranks through torch.distributed.run on this node (one process per
    GPU, RCCL over xGMI; rendezvous on 127.0.0.1) and pass its output / exit code through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VBX_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="train", choices=["train", "sample"])
    ap.add_argument("--gateloop", action="store_true", help="use_gateloop_layers=True (not a BASELINE config; off by default)")
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (weak scaling)")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--intervals", type=int, default=64, help="sample mode: midpoint intervals (NFE = 2x)")
    ap.add_argument("--attn-dropout", type=float, default=0.0, help="not a BASELINE config (the reference's default is 0): A/B only")
    ap.add_argument("--ff-dropout", type=float, default=0.0, help="not a BASELINE config: A/B only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sample", action="store_true", help="train mode: skip the 64-interval sample leg")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"], help="wire dtype of the gradient all-reduce (N > 1)")
    ap.add_argument("--bucket-mb", type=int, default=0, help="gradient all-reduce bucket size in MiB (0: the library default)")
    ap.add_argument("--adaln-exchange", default="auto", choices=["auto", "factors", "materialize"],
                    help="gradient of the adaLN projection weights (49 %% of the parameters, rank-B outer products): 'factors' keeps it in "
                         "factor form (N = 1: Adam expands it; N > 1: the factors are all-gathered instead of all-reducing the product: "
                         "half of the wire), 'materialize' is round 4's path")
    ap.add_argument("--grad-mode", default="allreduce", choices=["allreduce", "shard"],
                    help="N > 1: all-reduce + replicated optimizer (DDP semantics) or reduce-scatter + sharded clip / Adam + parameter all-gather")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("VBX_FORCE_DIST") == "1":  # VBX_FORCE_DIST=1: run the RCCL path at world size 1 too
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    vbx, vb, wrapper = build_model(args, dev)
    torch.manual_seed(1234 + rank)
    x = torch.randn(args.batch, args.frames, args.dim, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        from voicebox_pytorch_amd.dp import TrainStep

        kw = {"bucket_bytes": args.bucket_mb << 20} if args.bucket_mb > 0 else {}
        ts = TrainStep(wrapper, lr=3e-4, max_grad_norm=0.5, grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None,
                       grad_mode=args.grad_mode, adaln_grads=args.adaln_exchange, **kw)
        step = lambda: ts.step(x)
        units_per_step = args.batch * args.frames
        flops_per_step_per_gpu = 3.0 * fwd_flops_per_frame(args.dim, args.depth, args.heads, args.frames, 16) * units_per_step
        metric = "mel-frames/sec train-step (fwd+bwd+allreduce+clip+Adam)"
    else:
        steps_pts = args.intervals + 1
        step = lambda: wrapper.sample(cond=x, steps=steps_pts)
        units_per_step = args.batch * args.frames
        flops_per_step_per_gpu = 2.0 * args.intervals * fwd_flops_per_frame(args.dim, args.depth, args.heads, args.frames, 16) * units_per_step
        metric = f"mel-frames/sec ODE-sample ({args.intervals} midpoint intervals = {2 * args.intervals} NFE, hipGraph)"

    last = None
    for _ in range(args.warmup):
        last = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_val = float(last.float().mean()) if args.mode == "train" else None

    # ---- in-situ stage times (every rank runs the extra steps: they contain the collectives); sample mode replays a hipGraph,
    # whose launches do not pass through the runtime again, so its table comes from an eager 2-interval sample
    if args.mode == "train":
        rows = in_situ_stage_table(args, step, 3)
    else:
        # ONE stream, full batch (VBX_SAMPLE_SPLIT=1 for this sampler only): with the two concurrent half-batch streams of the timed
        # run the event brackets of the streams overlap and every launch works on B/2 -- such a table would credit full-batch FLOPs
        # to half-batch launches.  The wall-clock figure of the split run is `value` / sample.fwd_frac.
        prev = os.environ.get("VBX_SAMPLE_SPLIT")
        os.environ["VBX_SAMPLE_SPLIT"] = "1"
        try:
            rows = in_situ_stage_table(args, lambda: wrapper.sample(cond=x, steps=3, use_graph=False), 1)
        finally:
            if prev is None:
                del os.environ["VBX_SAMPLE_SPLIT"]
            else:
                os.environ["VBX_SAMPLE_SPLIT"] = prev
    barrier()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * units_per_step * args.steps / elapsed
        step_tf = flops_per_step_per_gpu / (elapsed / args.steps) / 1e12
        out = {
            "metric": metric, "value": round(value, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16/bf16 (fp16 forward operands, bf16 backward operands, fp32 accumulate/master)",
            "data": "synthetic",
            "config": {"workload": f"VoiceBox dim {args.dim} depth {args.depth} heads {args.heads} unconditional, "
                                   f"{args.batch}x{args.frames} frames per GPU, {args.mode}" + (", GateLoop layers" if args.gateloop else "")
                                   + (f", dropout attn {args.attn_dropout} ff {args.ff_dropout}" if args.attn_dropout or args.ff_dropout else ""),
                       "global_batch": world * args.batch, "seq_len": args.frames, "parallelism": f"dp{world}"},
            "step_tflops_per_gpu": round(step_tf, 1), "step_roofline_frac": round(step_tf / PEAK_MFMA_TFLOPS, 4),
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1,
        }
        if dist is not None:
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001  (diagnostic field only)
                out["rccl_version"] = f"unavailable ({type(e).__name__})"
            out["grad_comm"] = args.grad_comm
            out["grad_mode"] = args.grad_mode
            out["bucket_mb"] = args.bucket_mb
        if args.mode == "train":
            out["adaln_grads"] = "factors" if ts.adaln_factors_apply() else "materialize"
            out["wire_bytes_per_step"] = int(getattr(ts, "wire_bytes", 0) or 0)  # bytes this rank hands to collectives per step (0 at N = 1)
        if loss_val is not None:
            out["final_loss"] = round(loss_val, 5)
        # ---- roofline: the DOMINANT MFMA stage of the timed workload, timed in situ with HIP events on the launch stream
        mf = [r for r in rows if "frac" in r]
        if mf:
            top = mf[0]
            traffic, traffic_src = pmc_traffic(top["stage"])
            out["roofline"] = {"bound": "mfma", "kernel": top["stage"], "achieved": top["tflops"], "peak": PEAK_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": top["frac"], "traffic": traffic, "traffic_source": traffic_src,
                               "flops_per_launch": top["gflop_per_launch"] * 1e9, "us_per_launch": top["us_per_launch"],
                               "measured": "HIP events around the stage's launch(es) on the launch stream, in situ (vbx_prof_*), "
                                           + ("averaged over the layers of 3 extra steps" if args.mode == "train" else
                                              "one eager single-stream full-batch 2-interval sample (the timed run replays the captured "
                                              "interval graph(s): its wall clock is `value`)"),
                               "isolated_us": {k: isolated_gemm_us(args, dev, k) for k in ("fwd ff_in", "dgrad to_qkv")},
                               "mfma_us_per_step": round(sum(r["us_per_step"] for r in mf), 1),
                               "kernels": rows}
        if args.mode == "train" and world == 1 and not args.no_sample:
            # the other half of BASELINE.json's metric in the same invocation: 64 midpoint intervals = 128 NFE under hipGraph
            steps_pts = args.intervals + 1
            wrapper.sample(cond=x, steps=steps_pts)  # capture + warm-up
            torch.cuda.synchronize()
            dts = []
            for _ in range(3):  # the median of three runs: the first run after the capture (host-side work, an idle GPU) measured up to 5 % slow
                t0 = time.perf_counter()
                wrapper.sample(cond=x, steps=steps_pts)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[1]
            nfe = 2 * args.intervals
            fwd_flops = fwd_flops_per_frame(args.dim, args.depth, args.heads, args.frames, 16) * args.batch * args.frames
            out["sample"] = {"ms": round(dt * 1e3, 2), "frames_per_s": round(args.batch * args.frames / dt, 1), "nfe": nfe,
                             "ms_per_nfe": round(dt * 1e3 / nfe, 3), "fwd_frac": round(fwd_flops * nfe / dt / 1e12 / PEAK_MFMA_TFLOPS, 4),
                             "what": f"ConditionalFlowMatcherWrapper.sample(cond=(8,{args.frames},{args.dim}), steps={steps_pts}) "
                                     f"= {args.intervals} midpoint intervals under hipGraph (one stream at dim 512, where the weight-stationary GEMMs own whole "
                                     f"CUs; two concurrent half batches at other widths: solver.py), median of three timed runs after the capture run",
                             "runs_ms": [round(t * 1e3, 2) for t in dts]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
