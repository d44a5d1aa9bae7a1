"""Fixed-step midpoint ODE sampler replayed under hipGraph.

Replaces torchdiffeq.odeint(fn, y0, t, method='midpoint') (call site voicebox_pytorch.py:1295; torchdiffeq
is third-party, restated in oracle/ref_loader.py): per interval [t_i, t_{i+1}] of t = linspace(0,1,steps)
    f0 = fn(t_i, y);  f1 = fn(t_i + dt/2, y + f0*dt/2);  y <- y + dt*f1.
The grid values t_i, dt_i = t_{i+1}-t_i, dt_i/2 and t_i + dt_i/2 are computed on the host with the same
fp32 torch ops the oracle uses (linspace(0,1,64) has 8 distinct fp32 dt values -- SURVEY 8(c)) and live
in device tables; ONE interval (2 forwards + 2 axpys) is captured in a hipGraph and replayed steps-1 times,
a device counter selecting the table row, so no host scalar is baked into the graph.  Only the final
state is kept (the reference stacks the whole trajectory, voicebox_pytorch.py:1295-1296).

Concurrent halves.  Every kernel of a forward has a ramp, a drain and -- the GEMMs -- a VALU-bound epilogue during which the
matrix pipes idle (tools/native/gemm_trace.cpp); batch elements are independent in every kernel of the path.  So a batch of
B >= 4 (even) is integrated as TWO half-batches on two streams (the default except at dim 512, see MidpointSampler.__init__), each with its own engine (activation arena; the packed weights
are shared) and its own captured interval graph: one kernel stream fills the other's holes.  The two integrations never meet
before the end, and the second stream starts SPLIT_OFFSET_US late, so that different kernels of the two forwards overlap
(attention beside GEMMs) rather than the same ones.  Measured on the benchmark shape, 16 intervals: one stream 85.7 ms, two
joined branches of one graph 82.5, two free-running graphs 81.7, with the offset 80.1 (tools/sample_concurrent.py,
tools/sample_offset.py); results bit-identical to the single-stream run.  VBX_SAMPLE_SPLIT=1 restores the single stream (A/B).
Without a graph (use_graph=False) the halves run as fork / join branches per interval.
"""
import contextlib
import os

import torch

from . import _lib
from .engine import precise_enabled


SPLIT_OFFSET_US = float(os.environ.get("VBX_SAMPLE_OFFSET_US", "60"))  # start delay of the second (third, ...) half-batch stream; 30 .. 250 us measured equal


class _Part:
    """One concurrently integrated slice [lo, hi) of the batch: its engine and its views of the sampler's static buffers."""


class MidpointSampler:
    def __init__(self, voicebox, B, N, steps, use_graph=True, tokens=0, guided=False, split=None):
        assert steps >= 2, "need at least two time points"
        self.vb, self.B, self.N, self.steps = voicebox, B, N, steps
        if split is None:
            # Default: two concurrent half batches -- EXCEPT where the weight-stationary kernel serves to_qkv / FeedForward-in (dim 512,
            # csrc/gemm5.hip): it owns whole CUs, the half batches' launches cannot interleave with it, and ONE stream is then both
            # faster and deterministic (round 6, ten runs each on one box: 286.0 ms every run against 284-294, mostly 292; with the tiled
            # kernels 317.5 against 298-303 -- the split was a remedy for THEIR idle phases).  VBX_SAMPLE_SPLIT=1 / 2 overrides.
            env = os.environ.get("VBX_SAMPLE_SPLIT")
            if env is None:
                g5 = voicebox._cfg["D"] == 512 and os.environ.get("VBX_GEMM5", "1") != "0" and not precise_enabled()
                env = "1" if g5 else "2"
            if env not in ("1", "2"):
                raise ValueError(f"VBX_SAMPLE_SPLIT must be 1 or 2 (concurrent half batches are the only measured, tested split), got {env!r}")
            split = int(env)
        if split not in (1, 2):
            raise ValueError(f"MidpointSampler(split={split}): only 1 (one stream) and 2 (two concurrent half batches) are supported")
        if B < 4 or B % split:
            split = 1
        self.split = split
        Bp = B // split
        engines = [voicebox.engine(Bp, N, training=False)]
        for i in range(1, split):
            engines.append(voicebox.engine(Bp, N, training=False, slot=i, wpack_from=engines[0]))
        self.eng = engines[0]
        self.flat_gen = self.eng.fp.flat_gen  # the captured graph bakes in addresses inside this flat parameter buffer
        dev = self.eng.device
        D = voicebox._cfg.get("Din") or voicebox._cfg["D"]  # the ODE state lives in data space (dim_in)
        t = torch.linspace(0, 1, steps)  # host fp32, as the CPU oracle
        t0, dt = t[:-1], t[1:] - t[:-1]
        half = 0.5 * dt
        self.t_table = torch.stack((t0, t0 + half), dim=1).reshape(-1).contiguous().to(dev)   # [2*(steps-1)]
        self.c_table = torch.stack((half, dt), dim=1).reshape(-1).contiguous().to(dev)
        self.y = torch.zeros(B, N, D, device=dev)
        self.ymid = torch.zeros(B, N, D, device=dev)
        self.f = torch.zeros(B, N, D, device=dev)
        self.cond = torch.zeros(B, N, D, device=dev)
        self.cmask = torch.ones(B, N, dtype=torch.bool, device=dev)
        self.times = torch.zeros(B, device=dev)
        self.counters = torch.zeros(split, dtype=torch.int32, device=dev)  # one per part (each branch advances its own)
        # text-conditioned models: static token ids; classifier-free guidance (forward_with_cond_scale, :972-985) runs a
        # second, fully dropped evaluation (cond -> null_cond, ids -> null_cond_id) and mixes null + (logits - null) * scale
        self.tokens, self.guided = int(tokens), bool(guided)
        if self.tokens:
            self.ids = torch.zeros(B, self.tokens, dtype=torch.int64, device=dev)
            self.drop_all = torch.ones(B, dtype=torch.uint8, device=dev)
        if self.guided:
            assert self.tokens, "guidance needs a text-conditioned model"
            self.f_null = torch.zeros(B, N, D, device=dev)
            self.f_diff = torch.zeros(B, N, D, device=dev)
            self.g_table = torch.tensor([-1.0, 1.0], device=dev)  # [-1, cond_scale]
        self.parts = []
        for i, eng in enumerate(engines):
            p = _Part()
            p.eng, p.B = eng, Bp
            sl = slice(i * Bp, (i + 1) * Bp)
            p.y, p.ymid, p.f, p.cond, p.cmask, p.times = self.y[sl], self.ymid[sl], self.f[sl], self.cond[sl], self.cmask[sl], self.times[sl]
            p.counter = self.counters[i:i + 1]
            if self.tokens:
                p.ids, p.drop_all = self.ids[sl], self.drop_all[sl]
            if self.guided:
                p.f_null, p.f_diff = self.f_null[sl], self.f_diff[sl]
            self.parts.append(p)
        self.side_streams = [torch.cuda.Stream(device=dev) for _ in range(split - 1)]
        self.part_streams = [torch.cuda.Stream(device=dev) for _ in range(split)] if split > 1 else []
        # adaLN projections of every time point of the grid, evaluated once per weights version instead of once per function
        # evaluation (a 100 MB weight stream + the time MLP per forward at dim 512 / depth 12): every batch element shares the time.
        # VBX_SAMPLE_ADA_TABLE=0: A/B (per-forward projections, bit-identical results).
        self.use_ada_table = os.environ.get("VBX_SAMPLE_ADA_TABLE", "1") != "0"
        self.ada_tab, self.ada_key = None, None
        self.graph = None       # split == 1: one interval; split > 1: a list, one interval graph per part
        self.use_graph = use_graph
        self.nfe = 2 * (steps - 1) * (2 if self.guided else 1)

    def _bind(self, p, x, slot):
        # point the part's engine at the static buffers (x = y or ymid), prediction written to p.f; slot 0 / 1 = t_i / t_i + dt / 2
        p.eng.dropout_active = False  # sampling is eval (:1268) whatever mode a later forward of the same shape left on the engine
        ada = (self.ada_tab, p.counter, slot) if self.ada_tab is not None else None
        if not self.tokens:
            p.eng.forward(x, p.cond, p.cmask, p.times, pred_out=p.f, ada=ada)
            return
        vb = self.vb
        p.eng.forward(x, p.cond, p.cmask, p.times, pred_out=p.f, text=(p.ids, vb.null_cond_id, None, vb.null_cond), ada=ada)
        if self.guided:
            st, n = _lib.current_stream, p.f.numel()
            p.eng.forward(x, p.cond, p.cmask, p.times, pred_out=p.f_null,
                          text=(p.ids, vb.null_cond_id, p.drop_all, vb.null_cond), ada=ada)
            _lib.call("vbx_axpy_dev", p.f, p.f_null, self.g_table, 0, p.f_diff, n, st())   # logits - null
            _lib.call("vbx_axpy_dev", p.f_null, p.f_diff, self.g_table, 1, p.f, n, st())   # null + scale * diff

    def _interval_part(self, p):
        st = _lib.current_stream
        n = p.y.numel()
        if self.ada_tab is None:
            _lib.call("vbx_ode_set_time", p.times, p.B, self.t_table, p.counter, 0, st())
        self._bind(p, p.y, 0)
        _lib.call("vbx_axpy_ctr", p.y, p.f, self.c_table, p.counter, 0, p.ymid, n, st())
        if self.ada_tab is None:
            _lib.call("vbx_ode_set_time", p.times, p.B, self.t_table, p.counter, 1, st())
        self._bind(p, p.ymid, 1)
        _lib.call("vbx_axpy_ctr", p.y, p.f, self.c_table, p.counter, 1, p.y, n, st())
        _lib.call("vbx_counter_add", p.counter, 1, st())

    def _interval(self):
        # part 0 on the current stream, the others on side streams between a fork and a join: parallel branches under capture
        cur = torch.cuda.current_stream()
        for s in self.side_streams:
            s.wait_stream(cur)
        for p, s in zip(self.parts[1:], self.side_streams):
            with torch.cuda.stream(s):
                self._interval_part(p)
        self._interval_part(self.parts[0])
        for s in self.side_streams:
            cur.wait_stream(s)

    @contextlib.contextmanager
    def _cu_share(self):
        """The weight-stationary to_qkv / FeedForward-in kernel (csrc/gemm5.hip) owns whole CUs: with `split` concurrent parts every
        launch gets 1 / split of the chip (include/vbx.h vbx_gemm5_cu_limit; read at launch, so baked into the captured graphs) --
        the two parts' launches then run side by side instead of one behind the other."""
        if self.split == 1:
            yield
            return
        ncu = torch.cuda.get_device_properties(self.y.device).multi_processor_count
        share = int(os.environ.get("VBX_GEMM5_CUS", "0")) or max(ncu // self.split, 1)  # (VBX_GEMM5_CUS=<n>: A/B of the share)
        _lib.call("vbx_gemm5_cu_limit", share)
        try:
            yield
        finally:
            _lib.call("vbx_gemm5_cu_limit", 0)

    def _capture(self):
        with self._cu_share():
            self._capture_shared()

    def _capture_shared(self):
        # warm up on a side stream (one-time kernel attribute calls, weight packing), then capture one interval
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.counters.zero_()
            self._interval()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.split == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._interval()
            self.graph = g
        else:
            self.graph = []
            for p in self.parts:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._interval_part(p)
                self.graph.append(g)

    def _replay_parts(self):
        """steps-1 intervals of every part: each part's graph on its own stream, the streams never meet before the end."""
        cur = torch.cuda.current_stream()
        for s in self.part_streams:
            s.wait_stream(cur)
        for i, s in enumerate(self.part_streams[1:], 1):
            with torch.cuda.stream(s):
                _lib.call("vbx_stream_delay", SPLIT_OFFSET_US * i, _lib.current_stream())
        for _ in range(self.steps - 1):
            for g, s in zip(self.graph, self.part_streams):
                with torch.cuda.stream(s):
                    g.replay()
        for s in self.part_streams:
            cur.wait_stream(s)

    def run(self, y0, cond=None, cond_mask=None, cond_token_ids=None, cond_scale=1.0):
        # eval semantics of the reference: cond_mask None -> everything masked -> cond is zeroed (:1028-1035)
        if cond is not None:
            self.cond.copy_(cond)
        if cond_mask is not None:
            self.cmask.copy_(cond_mask.to(self.cmask.device))
        else:
            self.cmask.fill_(True)
        if self.tokens:
            self.ids.copy_(cond_token_ids.to(self.ids.device))
        if self.guided:
            self.g_table[1] = float(cond_scale)
        for p in self.parts:
            p.eng.bind_params()  # re-pack weights if they changed since the last call
        if self.use_ada_table and not self.vb._cfg.get("plain_norm"):
            key = self.eng.fp.weights_key()
            if self.ada_tab is None:
                self.ada_tab = self.eng.ada_table(self.t_table)  # allocated once: its address is baked into the captured graphs
                self.ada_key = key
            elif key != self.ada_key:
                self.ada_tab.copy_(self.eng.ada_table(self.t_table))
                self.ada_key = key
        if self.use_graph and self.graph is None:
            self._capture()
        self.y.copy_(y0)
        self.counters.zero_()
        if self.use_graph and self.split > 1:
            self._replay_parts()
        else:
            for _ in range(self.steps - 1):
                if self.use_graph:
                    self.graph.replay()
                else:
                    with self._cu_share():
                        self._interval()
        return self.y.clone()
