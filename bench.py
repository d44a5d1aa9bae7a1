"""Headline benchmark: mel-frames/sec of a full CFM train step (forward + backward + gradient all-reduce +
clip + Adam) of VoiceBox dim 512 / depth 12 / heads 16 on synthetic (B=8 per GPU, 1024 frames, dim 512)
-- BASELINE.json `metric`, configs[3] (the largest single-GPU configuration at N=1; weak scaling for N>1).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

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
                      condition_on_text=False, use_gateloop_layers=args.gateloop)
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


def dominant_kernel_roofline(args, dev):
    """Times the forward FeedForward-in GEMM (gemm_kernel<NT, EpiGEGLU, fp16>) at the benchmark shape: the largest
    single MFMA kernel of the layer (23.3 GF of 105 GF at dim 512), launched exactly as the runtime does."""
    from voicebox_pytorch_amd import _lib

    B, N, D, R = args.batch, args.frames, args.dim, 16
    M, F = B * (N + R), int(D * 4 * 2 / 3)
    Fp = (F + 63) // 64 * 64
    x = torch.randn(M, D, device=dev).half()
    w = (torch.randn(2 * Fp, D, device=dev) * D ** -0.5).half()
    bias = torch.zeros(2 * Fp, device=dev)
    g = torch.empty(M, Fp, dtype=torch.float16, device=dev)
    d = _lib.GemmDesc()
    d.mode, d.epilogue, d.M, d.N, d.K, d.lda, d.ldb, d.ldc = _lib.VBX_GEMM_NT, _lib.VBX_EPI_GEGLU, M, 2 * Fp, D, D, D, Fp
    d.A, d.B, d.C, d.bias, d.f16 = x.data_ptr(), w.data_ptr(), g.data_ptr(), bias.data_ptr(), 1
    lib = _lib.lib()
    st = _lib.current_stream()

    def launch():
        rc = lib.vbx_gemm(d, st)
        assert rc == 0

    sec = time_kernel(launch)
    # HBM bytes per launch of this kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of this same command; tools/pmc_summary.py applies the gfx950 read-side correction)
    traffic = None
    pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_bench_pmc_hbm.json")
    if os.path.exists(pmc) and (B, N, D) == (8, 1024, 512):
        for name, e in json.load(open(pmc)).items():
            if "EpiGEGLU" in name and "hbm_bytes_per_launch" in e:
                traffic = round(e["hbm_bytes_per_launch"])
    flops = 2.0 * M * D * 2 * F  # algorithmic (unpadded F) FLOPs per launch
    ach = flops / sec / 1e12
    return {"bound": "mfma", "kernel": "gemm_kernel<NT,EpiGEGLU,f16> (FeedForward-in + GEGLU)", "achieved": round(ach, 1),
            "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": traffic,
            "flops_per_launch": flops, "us_per_launch": round(sec * 1e6, 2)}


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's path, oracle/restate.py) on this box's host cores: one fwd+bwd CFM
    step on a bounded sample (batch 2 instead of 8, same frames/dim/depth)."""
    from oracle import restate

    # torch CPU matmuls scale poorly past a few dozen threads (256 threads: 300 s/step measured); cap and report it
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    cfg = restate.Cfg(dim=args.dim, depth=args.depth, heads=args.heads, dim_head=64)
    state = restate.init_state_dict(cfg, seed=0)
    p = {k: v.clone().requires_grad_(v.is_floating_point() and k != "null_cond") for k, v in state.items()}
    Bs = 1
    g = torch.Generator().manual_seed(0)
    x1, x0 = torch.randn(Bs, args.frames, args.dim, generator=g), torch.randn(Bs, args.frames, args.dim, generator=g)
    times, frac, rand = torch.rand(Bs, generator=g), 0.7 + 0.3 * torch.rand(Bs, generator=g), torch.rand(Bs, generator=g)
    best = None
    for _ in range(1):
        t0 = time.perf_counter()
        loss = restate.cfm_loss(p, cfg, x1, x0, times, frac, rand)
        loss.backward()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        for v in p.values():
            v.grad = None
    return {"value": round(Bs * args.frames / best, 1), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd (fp32, torch CPU, {cores} threads), batch {Bs} of {args.batch}, {args.frames} frames, "
                      f"dim {args.dim}, depth {args.depth}; 1 step; {best:.2f} s/step"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grad-comm", default="fp32", choices=["fp32", "bf16"], help="wire dtype of the gradient all-reduce (N > 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"

    vbx, vb, wrapper = build_model(args, dev)
    torch.manual_seed(1234 + rank)
    x = torch.randn(args.batch, args.frames, args.dim, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode == "train":
        from voicebox_pytorch_amd.dp import TrainStep

        ts = TrainStep(wrapper, lr=3e-4, max_grad_norm=0.5, grad_comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)
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
                                   f"{args.batch}x{args.frames} frames per GPU, {args.mode}" + (", GateLoop layers" if args.gateloop else ""),
                       "global_batch": world * args.batch, "seq_len": args.frames, "parallelism": f"dp{world}"},
            "step_tflops_per_gpu": round(step_tf, 1), "step_roofline_frac": round(step_tf / PEAK_MFMA_TFLOPS, 4),
        }
        if loss_val is not None:
            out["final_loss"] = round(loss_val, 5)
        out["roofline"] = dominant_kernel_roofline(args, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
