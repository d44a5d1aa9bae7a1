"""Host side of the native runtime: flat parameter storage, arenas, and the stage-level C-ABI calls
(vbx_model_forward / vbx_model_backward_*).  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import P, I, F

# enum mirrors of include/vbx.h
G_NAMES = ["SINW", "T1W", "T1B", "EMBW", "EMBB", "CONVW", "CONVB", "REG", "FNG", "PREDW", "CEMB"]
L_NAMES = ["G1W", "B1W", "G2W", "B2W", "G1B", "B1B", "G2B", "B2B", "QG", "KG", "QKVW", "OUTW", "FF1W", "FF1B", "FF2W", "FF2B",
           "GLG", "GLW", "GLLNW", "GLLNB", "N1G", "N2G", "SKW", "SKB"]
NG, NL = len(G_NAMES), len(L_NAMES)


class VbxModel(C.Structure):
    _fields_ = [("B", I), ("N", I), ("R", I), ("D", I), ("H", I), ("F", I), ("Th", I), ("L", I), ("ksize", I),
                ("qk_norm", I), ("attn_scale", F), ("training", I), ("params", P), ("grads", P), ("off", P),
                ("wpack", P), ("act", P), ("rot_cos", P), ("rot_sin", P), ("gateloop", I),
                ("stack_only", I), ("E", I), ("V1", I), ("plain_norm", I), ("attn_dropout", F), ("ff_dropout", F), ("Din", I),
                ("precise", I), ("wpack3", P), ("pscratch", P), ("unet", I), ("skip_scale", F), ("adaln_factors", I), ("defer_reduce", I), ("sq_partials", P)]


class VbxIO(C.Structure):
    _fields_ = [("x", P), ("cond", P), ("cond_mask", P), ("attn_mask", P), ("attn_mask_p", P), ("loss_mask", P),
                ("times", P), ("target", P), ("pred", P), ("loss", P), ("cond_ids", P), ("T", I), ("null_id", C.c_long),
                ("drop_mask", P), ("null_cond", P), ("dx", P), ("dcond", P), ("dropout", I), ("drop_seed", C.c_ulonglong),
                ("ada_table", P), ("ada_counter", P), ("ada_slot", I)]


class VbxAdamSeg(C.Structure):
    _fields_ = [("off", C.c_long), ("count", C.c_long), ("dst_bf16", P), ("dst_f16", P), ("dst_f32", P),
                ("cols", I), ("dst_ld", I), ("rowmap", I), ("F", I), ("block0", C.c_long)]


# ---- exact-operand ("precise") mode switch (include/vbx.h "precise mode", csrc/precise.hip)
import contextlib
import os

_precise = os.environ.get("VBX_PRECISE", "0") not in ("", "0")


def precise_enabled():
    return _precise


def set_precise(on):
    """Process-wide switch of the exact-operand forward: every forward matrix product to fp32 accuracy (hi/lo-split fp16 operands
    K-concatenated through the same MFMA tiles, fp32 attention) at ~3-4x the forward time.  Engines are cached per mode, so the
    switch may be flipped between calls.  Default: environment VBX_PRECISE (0)."""
    global _precise
    prev = _precise
    _precise = bool(on)
    return prev


@contextlib.contextmanager
def precise_mode(on=True):
    prev = set_precise(on)
    try:
        yield
    finally:
        set_precise(prev)


_runtime_protos_done = False


def _rt():
    global _runtime_protos_done
    l = _lib.lib()
    if not _runtime_protos_done:
        MP, IP = C.POINTER(VbxModel), C.POINTER(VbxIO)
        l.vbx_model_wpack_bytes.argtypes = [MP]
        l.vbx_model_wpack_bytes.restype = C.c_size_t
        l.vbx_model_act_bytes.argtypes = [MP]
        l.vbx_model_act_bytes.restype = C.c_size_t
        l.vbx_model_adam_segments.argtypes = [MP, C.c_long, C.POINTER(VbxAdamSeg), I, C.POINTER(C.c_long)]
        l.vbx_model_adam_segments.restype = I
        l.vbx_adam_step_packed.argtypes = [P, P, P, P, P, I, C.c_long, F, F, F, F, I, P, P]
        l.vbx_adam_step_packed.restype = I
        l.vbx_model_precise_wpack_bytes.argtypes = [MP]
        l.vbx_model_precise_wpack_bytes.restype = C.c_size_t
        l.vbx_model_precise_scratch_bytes.argtypes = [MP]
        l.vbx_model_precise_scratch_bytes.restype = C.c_size_t
        l.vbx_model_adaln_table.argtypes = [MP, P, I, P, P]
        l.vbx_model_adaln_table.restype = I
        l.vbx_model_adaln_factors.argtypes = [MP, C.POINTER(P), C.POINTER(P), C.POINTER(C.c_long), C.POINTER(P)]
        l.vbx_model_adaln_factors.restype = I
        l.vbx_model_sq_partials.argtypes = [MP, C.POINTER(C.c_long)]
        l.vbx_model_sq_partials.restype = C.c_long
        l.vbx_sumsq_adaln_factors.argtypes = [P, P, I, I, I, I, P, P]
        l.vbx_sumsq_adaln_factors.restype = I
        l.vbx_sumsq_ranges.argtypes = [P, C.POINTER(C.c_long), I, I, P, P, P]
        l.vbx_sumsq_ranges.restype = I
        l.vbx_adam_adaln_factors.argtypes = [P, P, P, C.POINTER(C.c_long), C.POINTER(P), P, P, I, I, I, I, F, F, F, F, I, P, P]
        l.vbx_adam_adaln_factors.restype = I
        l.vbx_adaln_expand_dw.argtypes = [P, P, P, I, I, I, I, P]
        l.vbx_adaln_expand_dw.restype = I
        for name, at in (("vbx_model_pack_weights", [MP, P]), ("vbx_model_pack_weights_precise", [MP, P]), ("vbx_model_forward", [MP, IP, P]),
                         ("vbx_model_backward_head", [MP, IP, P, P]), ("vbx_model_backward_layer", [MP, IP, I, P]),
                         ("vbx_model_backward_embed", [MP, IP, P])):
            fn = getattr(l, name)
            fn.argtypes = at
            fn.restype = I
        _runtime_protos_done = True
    return l


def _u8(mask):
    """bool mask -> uint8 storage for the kernels: a zero-copy view when possible (the .to(uint8) conversion is a kernel launch
    per mask per forward)."""
    if mask.dtype == torch.bool and mask.is_contiguous():
        return mask.view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


def _check(rc, what):
    if rc != 0:
        raise _lib.VbxError(f"{what} failed (rc={rc}): {_lib.lib().vbx_last_error().decode()}")


def rotary_tables(n_frames, n_registers, dim_head, theta, device):
    """cos/sin of the rotary angles, built with the reference's own torch ops on the host so the table
    is bit-identical to RotaryEmbedding.forward (voicebox_pytorch.py:173-191; registers sit at position
    -10000, :436-443).  freqs = cat(ang, ang) so only the first dim_head/2 columns are stored."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dim_head, 2).float() / dim_head))
    if n_registers > 0:
        pos = torch.cat((torch.full((n_registers,), -10000, dtype=torch.long), torch.arange(n_frames, dtype=torch.long)))
    else:
        pos = torch.arange(n_frames, dtype=torch.long)
    ang = torch.einsum("i,j->ij", pos.type_as(inv_freq), inv_freq)
    return ang.cos().contiguous().to(device), ang.sin().contiguous().to(device)


class FlatParams:
    """All trainable parameters of a VoiceBox as views into ONE flat fp32 buffer, ordered so that the
    backward pass completes gradients front-to-back (head, layers L-1..0, embed/time): contiguous
    ranges can be all-reduced while earlier layers are still computing."""

    def __init__(self, named_slots, depth):
        # named_slots: dict slot-name -> nn.Parameter ; slot names "PREDW", "L3.QKVW", ...
        order = ["PREDW", "FNG"]
        for l in reversed(range(depth)):
            order += [f"L{l}.{n}" for n in L_NAMES if f"L{l}.{n}" in named_slots]
        tail = ["EMBW", "EMBB", "CEMB", "CONVW", "CONVB", "REG", "SINW", "T1W", "T1B"]
        order += tail
        self.order = [s for s in order if s in named_slots]
        self.slots = named_slots
        self.depth = depth
        self.offsets = {}
        off = 0
        for s in self.order:
            self.offsets[s] = off
            off += named_slots[s].numel()
            off = (off + 63) // 64 * 64  # keep every slot 256-byte aligned (16-byte vector loads need 4)
        self.numel = off
        self.flat = None
        # weights epoch: bumped by every writer that goes around PyTorch's version counters (the native Adam writes the flat
        # buffer through a raw pointer) and by every re-flatten; flat_gen counts re-flattens only (a captured hipGraph has the
        # old buffer's addresses baked in and must be dropped when it changes).
        self.epoch = 0
        self.flat_gen = 0
        # stage boundaries (in floats) for bucketed gradient exchange: [head][layer L-1]...[layer 0][embed]
        self.stage_ranges = []
        first_layer_slot = lambda l: next(s for s in self.order if s.startswith(f"L{l}."))
        first_tail = next((self.offsets[t] for t in tail if t in self.offsets), off)  # a bare Transformer has REG at most
        bounds = [0] + [self.offsets[first_layer_slot(l)] for l in reversed(range(depth))] + [first_tail, off]
        self.stage_ranges = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]

    def is_current(self):
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        for s in self.order:
            p = self.slots[s]
            if p.data_ptr() != base + 4 * self.offsets[s] or p.dtype != torch.float32:
                return False
        return True

    def flatten(self):
        dev = self.slots[self.order[0]].device
        # may be reached from sample() (inference_mode): the flat buffer must be a normal tensor (version counter)
        with torch.inference_mode(False), torch.no_grad():
            flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            for s in self.order:
                p = self.slots[s]
                o = self.offsets[s]
                view = flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.to(device=dev, dtype=torch.float32))
                p.data = view
        self.flat = flat
        self.epoch += 1
        self.flat_gen += 1
        return flat

    def bump(self):
        """Call after writing parameter values behind PyTorch's back (raw-pointer kernels, collectives into `flat`) -- and after
        writes through `p.data` (p.data.copy_(), p.data.mul_(), EMA weight swaps): `.data` has its own version counter, so such a
        write changes neither p._version nor flat._version and the engines (and the hipGraph samplers that hold them) would keep
        serving the previous fp16 / bf16 packed copies.  Public spelling: VoiceBox.mark_weights_dirty()."""
        self.epoch += 1

    def weights_key(self):
        """Changes whenever any parameter value may have changed.  The nn.Parameters are `p.data = view` tensors with their OWN
        version counters: in-place updates through them (torch.optim.*.step, load_state_dict, p.copy_) bump p._version and never
        flat._version, so the key sums the per-parameter counters; native writers bump `epoch`."""
        ver = 0
        for s in self.order:
            ver += self.slots[s]._version
        return (self.flat.data_ptr(), self.flat._version, self.epoch, ver)

    def offset_table(self):
        tab = (C.c_long * (NG + self.depth * NL))()
        for i, n in enumerate(G_NAMES):
            tab[i] = self.offsets.get(n, 0)
        for l in range(self.depth):
            for j, n in enumerate(L_NAMES):
                tab[NG + l * NL + j] = self.offsets.get(f"L{l}.{n}", 0)
        return tab

    def grad_views(self, gflat):
        return [gflat[self.offsets[s]:self.offsets[s] + self.slots[s].numel()].view(self.slots[s].shape) for s in self.order]


class Engine:
    """One (batch, frames, training) configuration of a VoiceBox on one GPU: owns the packed-weight and
    activation arenas and issues the native stage calls on the current HIP stream."""

    def __init__(self, cfg, flat: FlatParams, B, N, training, device, wpack_from=None, precise=False):
        """wpack_from: another Engine of the same model whose packed-weight arena this one shares (the arena depends on the
        architecture only): the concurrent half-batch engines of the sampler keep ONE copy of the operand weights.
        precise: the exact-operand forward (csrc/precise.hip): two more arenas (split weights, fp32 intermediates)."""
        self.cfg, self.fp, self.B, self.N, self.training, self.device = cfg, flat, B, N, bool(training), device
        self.precise = bool(precise)
        self.wpack_owner = wpack_from
        _lib.call("vbx_check_device", device.index if device.index is not None else torch.cuda.current_device())
        l = _rt()
        m = VbxModel()
        m.B, m.N, m.R, m.D, m.H, m.F, m.Th, m.L, m.ksize = (B, N, cfg["R"], cfg["D"], cfg["H"], cfg["F"], cfg["Th"],
                                                            cfg["L"], cfg["ksize"])
        m.qk_norm = 1 if cfg["qk_norm"] else 0
        m.attn_scale = float(cfg["attn_scale"])
        m.training = 1 if training else 0
        m.gateloop = 1 if cfg.get("gateloop") else 0
        m.stack_only = 1 if cfg.get("stack_only") else 0
        m.E, m.V1 = int(cfg.get("E", 0)), int(cfg.get("V1", 0))
        m.plain_norm = 1 if cfg.get("plain_norm") else 0
        m.attn_dropout = float(cfg.get("attn_dropout", 0.))
        m.ff_dropout = float(cfg.get("ff_dropout", 0.))
        m.Din = int(cfg.get("Din", 0) or 0)  # data width (dim_in); 0 = D
        m.unet = 1 if cfg.get("unet") else 0
        m.skip_scale = float(cfg.get("skip_scale", 2 ** -0.5))
        self.Din = m.Din or cfg["D"]
        self.has_dropout = m.attn_dropout > 0. or m.ff_dropout > 0.
        self.dropout_active = False  # nn.Dropout semantics: the owning module sets this to its .training flag before a forward
        self.off_table = flat.offset_table()
        m.off = C.cast(self.off_table, P)
        self.rot_cos, self.rot_sin = rotary_tables(N, cfg["R"], 64, cfg["theta"], device)
        m.rot_cos, m.rot_sin = self.rot_cos.data_ptr(), self.rot_sin.data_ptr()
        self.m = m
        if wpack_from is not None:
            assert wpack_from.wpack.numel() == l.vbx_model_wpack_bytes(C.byref(m)) and wpack_from.fp is flat
            self.wpack = wpack_from.wpack
        else:
            self.wpack = torch.empty(l.vbx_model_wpack_bytes(C.byref(m)), dtype=torch.uint8, device=device)
        self.act = torch.empty(l.vbx_model_act_bytes(C.byref(m)), dtype=torch.uint8, device=device)
        m.wpack, m.act = self.wpack.data_ptr(), self.act.data_ptr()
        self.packed3_version = None
        if self.precise:
            if wpack_from is not None:
                assert wpack_from.precise
                self.wpack3 = wpack_from.wpack3
            else:
                self.wpack3 = torch.empty(l.vbx_model_precise_wpack_bytes(C.byref(m)), dtype=torch.uint8, device=device)
            self.pscratch = torch.empty(l.vbx_model_precise_scratch_bytes(C.byref(m)), dtype=torch.uint8, device=device)
            m.precise, m.wpack3, m.pscratch = 1, self.wpack3.data_ptr(), self.pscratch.data_ptr()
        self.packed_version = None
        self.io = VbxIO()
        self._keep = None
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        self.generation = 0

    # -- weights
    def bind_params(self):
        flat = self.fp.flat
        self.m.params = flat.data_ptr()
        if self.wpack_owner is not None:  # shared arena: its owner packs
            self.wpack_owner.bind_params()
            return
        key = self.fp.weights_key()
        if key != self.packed_version:
            _check(_rt().vbx_model_pack_weights(C.byref(self.m), _lib.current_stream()), "vbx_model_pack_weights")
            self.packed_version = key
        if self.precise and key != self.packed3_version:  # the fused Adam refreshes the plain copies only: repack the split ones
            _check(_rt().vbx_model_pack_weights_precise(C.byref(self.m), _lib.current_stream()), "vbx_model_pack_weights_precise")
            self.packed3_version = key

    # -- optimizer: Adam over the flat buffers that also refreshes this engine's packed operand copies
    # -- adaLN weight gradients in factor form (include/vbx.h "FACTOR form"; dp.TrainStep decides when)
    def supports_adaln_factors(self):
        c = self.cfg
        return bool(self.training and not c.get("plain_norm") and not c.get("stack_only") and c["L"] <= 32)

    def adaln_factor_info(self):
        """(dada ptr, temb ptr, [w_off], [dst_f16 ptr], J4, Th) of this arena's factor-form adaLN weight gradients."""
        if getattr(self, "_adaln_info", None) is None:
            L = self.cfg["L"]
            dada, temb = P(), P()
            woff, dst = (C.c_long * L)(), (P * L)()
            _check(0 if _rt().vbx_model_adaln_factors(C.byref(self.m), C.byref(dada), C.byref(temb), woff, dst) == L else -1,
                   "vbx_model_adaln_factors")
            self._adaln_info = (dada.value, temb.value, woff, dst, 4 * self.cfg["D"], self.cfg["Th"])
        return self._adaln_info

    def adaln_factor_tensors(self):
        """(dada [L, B, 4 D], temb [B, Th]) as fp32 views of this arena (valid between a backward and the next forward)."""
        dada_p, temb_p, _, _, J4, Th = self.adaln_factor_info()
        L, B = self.cfg["L"], self.B

        def view(ptr, shape):
            off, n = ptr - self.act.data_ptr(), 4
            for d in shape:
                n *= d
            return self.act[off:off + n].view(torch.float32).view(*shape)

        return view(dada_p, (L, B, J4)), view(temb_p, (B, Th))

    def adaln_factor_ranges(self):
        """Flat ranges [lo, hi) of the adaLN projection weight blocks (one per layer, ascending)."""
        _, _, woff, _, J4, Th = self.adaln_factor_info()
        return sorted((int(o), int(o) + J4 * Th) for o in woff)

    @staticmethod
    def complement_ranges(n, blocks):
        """blocks: [lo, hi) ranges inside [0, n) in any order -> the gaps between them as a flat list [lo0, hi0, lo1, hi1, ...]"""
        rest, cur = [], 0
        for lo, hi in sorted(blocks):
            if lo > cur:
                rest += [cur, lo]
            cur = max(cur, hi)
        if cur < n:
            rest += [cur, n]
        return rest

    SUMSQ_MAX_RANGES = 64  # vbx_sumsq_ranges

    def sq_partials_info(self):
        """(floats, [(lo, hi), ...]): size of the slab-reduce sum-of-squares partials (vbx_model.sq_partials) and the flat gradient
        ranges they cover -- every layer's to_qkv / to_out / FeedForward weights; (0, []) when this configuration does not serve them."""
        if getattr(self, "_sq_info", None) is None:
            L = self.cfg["L"]
            arr = (C.c_long * (8 * L))()
            n = int(_rt().vbx_model_sq_partials(C.byref(self.m), arr))
            cov = sorted((int(arr[2 * i]), int(arr[2 * i + 1])) for i in range(4 * L)) if n > 0 else []
            if n > 0 and self.supports_adaln_factors():
                # the remaining small tensors are read by ONE vbx_sumsq_ranges launch: a deep model has more gaps than it takes
                gaps = self.complement_ranges(self.fp.flat.numel(), list(self.adaln_factor_ranges()) + cov)
                if len(gaps) // 2 > self.SUMSQ_MAX_RANGES:
                    n, cov = 0, []
            self._sq_info = (n, cov)
        return self._sq_info

    def sumsq_scratch_floats(self, sq_fold=False):
        """floats the scratch of sumsq_with_adaln_factors must hold: 1024 block partials + L * B * B factor terms (+ the slab partials)"""
        return 1024 + self.cfg["L"] * self.B * self.B + (self.sq_partials_info()[0] if sq_fold else 0)

    def sq_partials_ptr(self, scratch):
        """where in `scratch` the backward must leave the slab partials for sumsq_with_adaln_factors(..., sq_fold=True)"""
        return scratch.data_ptr() + 4 * (1024 + self.cfg["L"] * self.B * self.B)

    def sumsq_with_adaln_factors(self, gflat, out, scratch, sq_fold=False):
        """out[0] = sum of squares of the gradient whose adaLN weight blocks are in factor form: the flat buffer minus those blocks
        (never written in that mode) plus |dada_l^T . temb|_F^2 per layer from the factors.  scratch: >= 1024 + L * B * B floats.
        sq_fold: the last backward ran with sq_partials = sq_partials_ptr(scratch) -- the big weight matrices' terms are already
        there and only the small tensors are read again (sumsq_scratch_floats(True) floats of scratch)."""
        l = _rt()
        dada, temb, woff, dst, J4, Th = self.adaln_factor_info()
        L, n = self.cfg["L"], gflat.numel()
        st = _lib.current_stream()
        nterms = L * self.B * self.B  # (dada_l[b] . dada_l[b']) (temb[b] . temb[b']) for every (l, b, b')
        nsq, covered = self.sq_partials_info() if sq_fold else (0, [])
        assert scratch.numel() >= 1024 + nterms + nsq, "sumsq scratch too small for the factor terms"
        _check(l.vbx_sumsq_adaln_factors(dada, temb, L, self.B, J4, Th, scratch.data_ptr() + 4 * 1024, st), "vbx_sumsq_adaln_factors")
        cache = self.__dict__.setdefault("_rest_ranges_by_mode", {})
        if bool(nsq) not in cache:
            rest = self.complement_ranges(n, list(self.adaln_factor_ranges()) + list(covered))
            assert len(rest) // 2 <= self.SUMSQ_MAX_RANGES, "too many gradient ranges for vbx_sumsq_ranges"
            cache[bool(nsq)] = ((C.c_long * len(rest))(*rest), len(rest) // 2)
        arr, nr = cache[bool(nsq)]
        _check(l.vbx_sumsq_ranges(gflat.data_ptr(), arr, nr, nterms + nsq, out.data_ptr(), scratch.data_ptr(), st), "vbx_sumsq_ranges")

    # -- optimizer: Adam over the flat buffers that also refreshes this engine's packed operand copies
    def adam_step_packed(self, gflat, m, v, lr, beta1, beta2, eps, step, gscale, adaln_factors=False):
        """adaln_factors=True: the adaLN weight blocks are updated from their factor-form gradients (the last backward ran with
        adaln_factors) by a second launch; the fused launch leaves them out."""
        l = _rt()
        flat = self.fp.flat
        self.bind_params()  # the packed arena must be current before it is updated incrementally
        cache = self.__dict__.setdefault("_adam_segs_by_mode", {})
        if bool(adaln_factors) not in cache:
            self.m.adaln_factors = int(bool(adaln_factors))
            try:
                nblocks = C.c_long(0)
                n = l.vbx_model_adam_segments(C.byref(self.m), flat.numel(), None, 0, C.byref(nblocks))
                if n <= 0:
                    _check(n if n < 0 else -1, "vbx_model_adam_segments")
                tab = (VbxAdamSeg * n)()
                n2 = l.vbx_model_adam_segments(C.byref(self.m), flat.numel(), tab, n, C.byref(nblocks))
                assert n2 == n
            finally:
                self.m.adaln_factors = 0
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
            cache[bool(adaln_factors)] = (raw, n, nblocks.value)
        raw, n, nblocks = cache[bool(adaln_factors)]
        gs = gscale.data_ptr() if gscale is not None else None
        _check(l.vbx_adam_step_packed(flat.data_ptr(), gflat.data_ptr(), m.data_ptr(), v.data_ptr(), raw.data_ptr(), n, nblocks,
                                      float(lr), float(beta1), float(beta2), float(eps), int(step), gs, _lib.current_stream()),
               "vbx_adam_step_packed")
        if adaln_factors:
            dada, temb, woff, dst, J4, Th = self.adaln_factor_info()
            _check(l.vbx_adam_adaln_factors(flat.data_ptr(), m.data_ptr(), v.data_ptr(), woff, dst, dada, temb, self.cfg["L"], self.B, J4,
                                            Th, float(lr), float(beta1), float(beta2), float(eps), int(step), gs, _lib.current_stream()),
                   "vbx_adam_adaln_factors")
        self.fp.bump()  # the flat buffer was written through a raw pointer: every other engine must repack
        self.packed_version = self.fp.weights_key()  # this engine's operand copies were refreshed in the same pass

    # -- forward
    def forward(self, x, cond, cond_mask, times, attn_mask=None, target=None, loss_mask=None, pred_out=None, text=None, ada=None):
        """All tensors on self.device, fp32 contiguous / bool.  Returns the loss tensor (1,) if target is given,
        else the prediction (B,N,D).  text (text-conditioned models): (ids int64 (B,T), null_id, drop_mask bool (B,) or None,
        null_cond fp32 (D,)).  ada (inference, the sampler): (table fp32 [2 * intervals, L, 4 * D], counter int32 [1], slot) -- the
        adaLN projections of every time point of the ODE grid, precomputed; `times` is then not read."""
        self.bind_params()
        B, N, D = self.B, self.N, self.Din
        assert x.shape == (B, N, D) and cond.shape == (B, N, D), (tuple(x.shape), tuple(cond.shape), (B, N, D))
        x, cond = x.contiguous(), cond.contiguous()
        cm = _u8(cond_mask)
        am = amp = lm = None
        if attn_mask is not None:
            am = _u8(attn_mask)
            R = self.cfg["R"]
            amp = torch.cat((torch.ones(B, R, dtype=torch.uint8, device=am.device), am), dim=1).contiguous() if R else am
        io = self.io
        io.x, io.cond, io.cond_mask = x.data_ptr(), cond.data_ptr(), cm.data_ptr()
        io.attn_mask = am.data_ptr() if am is not None else None
        io.attn_mask_p = amp.data_ptr() if amp is not None else None
        io.times = times.data_ptr()
        if ada is not None:
            assert target is None and not self.training
            io.ada_table, io.ada_counter, io.ada_slot = ada[0].data_ptr(), ada[1].data_ptr(), int(ada[2])
        else:
            io.ada_table = io.ada_counter = None
            io.ada_slot = 0
        if target is not None:
            target = target.contiguous()
            lm = _u8(loss_mask)
            io.target, io.loss_mask, io.loss = target.data_ptr(), lm.data_ptr(), self.loss.data_ptr()
            pred = None
            io.pred = None
        else:
            io.target = io.loss_mask = io.loss = None
            pred = pred_out if pred_out is not None else torch.empty(B, N, D, dtype=torch.float32, device=self.device)
            io.pred = pred.data_ptr()
        if self.cfg.get("E", 0):
            ids, null_id, drop, null_cond = text
            ids = ids.to(torch.int64).contiguous()
            drop8 = _u8(drop) if drop is not None else None
            null_cond = null_cond.detach().to(torch.float32).contiguous()
            io.cond_ids, io.T, io.null_id = ids.data_ptr(), int(ids.shape[1]), int(null_id)
            io.drop_mask = drop8.data_ptr() if drop8 is not None else None
            io.null_cond = null_cond.data_ptr()
            text = (ids, drop8, null_cond)
        self._keep = (x, cond, cm, am, amp, lm, times, target, pred, text, ada)  # keep inputs alive until backward
        self.generation += 1
        self._draw_dropout_seed()
        _check(_rt().vbx_model_forward(C.byref(self.m), C.byref(io), _lib.current_stream()), "vbx_model_forward")
        return self.loss if target is not None else pred

    def ada_table(self, times):
        """adaLN projections [len(times), L, 4 * D] of the given time points (fp32, device): the same kernels the forward runs
        (vbx_time_embed_fwd + vbx_adaln_proj_fwd: every (time, output) pair is an independent dot product, so a row of the table is
        bit-identical to what a forward at that time computes for each of its batch rows), 16 time points per launch."""
        self.bind_params()
        cfg, dev = self.cfg, self.device
        D, Th, L = cfg["D"], cfg["Th"], cfg["L"]
        off = self.fp.offsets
        flat = self.fp.flat
        self.m.params = flat.data_ptr()
        times = times.to(dev, torch.float32).contiguous()
        T = times.numel()
        out = torch.empty(T, L, 4 * D, dtype=torch.float32, device=dev)
        st = _lib.current_stream
        sinw, t1w, t1b = (flat[off[k]:] for k in ("SINW", "T1W", "T1B"))
        for t0 in range(0, T, 16):
            n = min(16, T - t0)
            four = torch.empty(n, D, device=dev)
            pre = torch.empty(n, Th, device=dev)
            temb = torch.empty(n, Th, device=dev)
            _lib.call("vbx_time_embed_fwd", times[t0:t0 + n], sinw, t1w, t1b, four, pre, temb, n, D, Th, st())
            tmp = torch.empty(L, n, 4 * D, device=dev)
            _check(_rt().vbx_model_adaln_table(C.byref(self.m), temb.data_ptr(), n, tmp.data_ptr(), st()), "vbx_model_adaln_table")
            out[t0:t0 + n] = tmp.permute(1, 0, 2)
        return out

    def _draw_dropout_seed(self):
        """One Philox key per training forward, drawn from torch's CPU generator (so torch.manual_seed reproduces a run); the
        backward calls reuse it through self.io (masks are a pure function of seed, layer and element index)."""
        self.io.dropout = 0
        if self.has_dropout and self.dropout_active:
            self.io.dropout = 1
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # data-parallel ranks seeded alike (torch.manual_seed(s) everywhere) must still drop different entries
                seed ^= (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & (2 ** 62 - 1)
            self.io.drop_seed = seed

    # -- standalone Transformer.forward / backward (vbx_model.stack_only)
    def forward_stack(self, x, cond=None, attn_mask=None):
        """x (B,N,D) fp32, cond (B,Th) fp32 or None (plain RMSNorm), attn_mask (B,N) bool or None -> (B,N,D) fp32."""
        self.bind_params()
        B, N, D, R = self.B, self.N, self.cfg["D"], self.cfg["R"]
        x = x.contiguous()
        cond = cond.contiguous() if cond is not None else None
        am = amp = None
        if attn_mask is not None:
            am = _u8(attn_mask)
            amp = torch.cat((torch.ones(B, R, dtype=torch.uint8, device=am.device), am), dim=1).contiguous() if R else am
        out = torch.empty(B, N, D, dtype=torch.float32, device=self.device)
        io = self.io
        io.x, io.cond = x.data_ptr(), (cond.data_ptr() if cond is not None else None)
        io.cond_mask = io.times = io.target = io.loss_mask = io.loss = io.dx = io.dcond = None
        io.attn_mask = am.data_ptr() if am is not None else None
        io.attn_mask_p = amp.data_ptr() if amp is not None else None
        io.pred = out.data_ptr()
        self._keep = (x, cond, am, amp, out)
        self.generation += 1
        self._draw_dropout_seed()
        _check(_rt().vbx_model_forward(C.byref(self.m), C.byref(io), _lib.current_stream()), "vbx_model_forward")
        return out

    def backward_stack(self, gflat, dout):
        """dout (B,N,D) fp32 -> (dx (B,N,D), dcond (B,Th) or None); parameter gradients land in gflat."""
        assert self.training
        dout = dout.to(torch.float32).contiguous()
        dx = torch.empty_like(dout)
        dcond = None if self.cfg.get("plain_norm") else torch.empty(self.B, self.cfg["Th"], dtype=torch.float32, device=self.device)
        io = self.io
        io.target, io.dx, io.dcond = dout.data_ptr(), dx.data_ptr(), (dcond.data_ptr() if dcond is not None else None)
        self.m.grads = gflat.data_ptr()
        st, l = _lib.current_stream(), _rt()
        _check(l.vbx_model_backward_head(C.byref(self.m), C.byref(io), None, st), "vbx_model_backward_head")
        for layer in reversed(range(self.cfg["L"])):
            _check(l.vbx_model_backward_layer(C.byref(self.m), C.byref(io), layer, st), "vbx_model_backward_layer")
        _check(l.vbx_model_backward_embed(C.byref(self.m), C.byref(io), st), "vbx_model_backward_embed")
        self.m.grads = None
        io.target = io.dx = io.dcond = None
        return dx, dcond

    def replay_forward(self):
        """Re-issue the last forward with the same (static) buffers -- the body of the captured ODE step."""
        _check(_rt().vbx_model_forward(C.byref(self.m), C.byref(self.io), _lib.current_stream()), "vbx_model_forward")

    def debug_tensor(self, name, layer, shape, dtype):
        """tests/debug: view of a named arena tensor."""
        l = _rt()
        l.vbx_model_debug_ptr.argtypes = [C.POINTER(VbxModel), C.c_char_p, I]
        l.vbx_model_debug_ptr.restype = C.c_void_p
        p = l.vbx_model_debug_ptr(C.byref(self.m), name.encode(), layer)
        if not p:
            raise KeyError(name)
        off = p - self.act.data_ptr()
        n = int(torch.tensor(shape).prod()) * torch.empty(0, dtype=dtype).element_size()
        return self.act[off:off + n].view(dtype).view(*shape)

    # -- backward, stage by stage; `on_stage(i, (lo, hi))` fires after the gradients in flat range [lo,hi) are final
    def backward(self, gflat, gscale=None, on_stage=None, adaln_factors=False, sq_partials=None):
        """adaln_factors=True: the adaLN projection WEIGHT gradients are not written into gflat -- they stay in factor form in this
        arena (adaln_factor_info) until the next forward; the caller's optimizer / exchange must take them from there.
        sq_partials: device address of sq_partials_info()[0] floats -- the slab reduce leaves the big weight matrices' sums of
        squares there (the caller takes the gradient norm of THIS backward from them: no accumulation / exchange in between)."""
        assert self.training
        assert not adaln_factors or self.supports_adaln_factors()
        self.m.adaln_factors = int(bool(adaln_factors))
        self.m.defer_reduce = int(on_stage is None)  # nobody reads a gradient before the last stage: one reduce for all layers
        self.m.sq_partials = sq_partials
        try:
            return self._backward(gflat, gscale, on_stage)
        finally:
            self.m.adaln_factors = 0
            self.m.defer_reduce = 0
            self.m.sq_partials = None

    def _backward(self, gflat, gscale, on_stage):
        self.m.grads = gflat.data_ptr()
        st = _lib.current_stream()
        l = _rt()
        ranges = self.fp.stage_ranges
        _check(l.vbx_model_backward_head(C.byref(self.m), C.byref(self.io), gscale.data_ptr() if gscale is not None else None, st),
               "vbx_model_backward_head")
        if on_stage:
            on_stage(0, ranges[0])
        L = self.cfg["L"]
        for i, layer in enumerate(reversed(range(L))):
            _check(l.vbx_model_backward_layer(C.byref(self.m), C.byref(self.io), layer, st), "vbx_model_backward_layer")
            if on_stage:
                on_stage(1 + i, ranges[1 + i])
        _check(l.vbx_model_backward_embed(C.byref(self.m), C.byref(self.io), st), "vbx_model_backward_embed")
        if on_stage:
            on_stage(1 + L, ranges[1 + L])
        self.m.grads = None
