"""Fixed-step midpoint ODE sampler replayed under hipGraph.

Replaces torchdiffeq.odeint(fn, y0, t, method='midpoint') (call site voicebox_pytorch.py:1295; torchdiffeq
is third-party, restated in oracle/ref_loader.py): per interval [t_i, t_{i+1}] of t = linspace(0,1,steps)
    f0 = fn(t_i, y);  f1 = fn(t_i + dt/2, y + f0*dt/2);  y <- y + dt*f1.
The grid values t_i, dt_i = t_{i+1}-t_i, dt_i/2 and t_i + dt_i/2 are computed on the host with the same
fp32 torch ops the oracle uses (linspace(0,1,64) has 8 distinct fp32 dt values -- SURVEY 8(c)) and live
in device tables; ONE interval (2 forwards + 2 axpys) is captured in a hipGraph and replayed steps-1 times,
a device counter selecting the table row, so no host scalar is baked into the graph.  Only the final
state is kept (the reference stacks the whole trajectory, voicebox_pytorch.py:1295-1296).
"""
import torch

from . import _lib


class MidpointSampler:
    def __init__(self, voicebox, B, N, steps, use_graph=True, tokens=0, guided=False):
        assert steps >= 2, "need at least two time points"
        self.vb, self.B, self.N, self.steps = voicebox, B, N, steps
        self.eng = voicebox.engine(B, N, training=False)
        self.flat_gen = self.eng.fp.flat_gen  # the captured graph bakes in addresses inside this flat parameter buffer
        dev = self.eng.device
        D = voicebox._cfg["D"]
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
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev)
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
        self.graph = None
        self.use_graph = use_graph
        self.nfe = 2 * (steps - 1) * (2 if self.guided else 1)

    def _bind(self, x):
        # point the engine's io at the static buffers (x = y or ymid), prediction written to self.f
        if not self.tokens:
            self.eng.forward(x, self.cond, self.cmask, self.times, pred_out=self.f)
            return
        vb = self.vb
        self.eng.forward(x, self.cond, self.cmask, self.times, pred_out=self.f, text=(self.ids, vb.null_cond_id, None, vb.null_cond))
        if self.guided:
            st, n = _lib.current_stream, self.f.numel()
            self.eng.forward(x, self.cond, self.cmask, self.times, pred_out=self.f_null,
                             text=(self.ids, vb.null_cond_id, self.drop_all, vb.null_cond))
            _lib.call("vbx_axpy_dev", self.f, self.f_null, self.g_table, 0, self.f_diff, n, st())   # logits - null
            _lib.call("vbx_axpy_dev", self.f_null, self.f_diff, self.g_table, 1, self.f, n, st())   # null + scale * diff

    def _interval(self):
        st = _lib.current_stream
        n = self.y.numel()
        _lib.call("vbx_ode_set_time", self.times, self.B, self.t_table, self.counter, 0, st())
        self._bind(self.y)
        _lib.call("vbx_axpy_ctr", self.y, self.f, self.c_table, self.counter, 0, self.ymid, n, st())
        _lib.call("vbx_ode_set_time", self.times, self.B, self.t_table, self.counter, 1, st())
        self._bind(self.ymid)
        _lib.call("vbx_axpy_ctr", self.y, self.f, self.c_table, self.counter, 1, self.y, n, st())
        _lib.call("vbx_counter_add", self.counter, 1, st())

    def _capture(self):
        # warm up on a side stream (one-time kernel attribute calls, weight packing), then capture one interval
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.counter.zero_()
            self._interval()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._interval()
        self.graph = g

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
        self.eng.bind_params()  # re-pack weights if they changed since the last call
        if self.use_graph and self.graph is None:
            self._capture()
        self.y.copy_(y0)
        self.counter.zero_()
        for _ in range(self.steps - 1):
            if self.use_graph:
                self.graph.replay()
            else:
                self._interval()
        return self.y.clone()
