"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the UNMODIFIED reference package from /root/reference (read-only, only
present in the build container, NOT on the GPU box) so that

  * oracle/restate.py can be validated against the real thing, and
  * tests/golden/make_golden.py can generate the committed fixtures.

The reference cannot be imported as-is here: voicebox_pytorch.py:13-36 and
__init__.py:13-15 import nine third-party packages that are not installed
(torchode, torchdiffeq, beartype, naturalspeech2_pytorch, audiolm_pytorch,
spear_tts_pytorch, gateloop_transformer, torchaudio, vocos).  We inject
`sys.modules` stubs for them.  Every stub replaces a symbol that the
unconditional hot path never *calls*, with two exceptions that carry real
arithmetic and are therefore restated here:

  * torchdiffeq.odeint(method='midpoint')  (call site voicebox_pytorch.py:1295)
    torchdiffeq is unpinned in setup.py:27 and its source is absent, so this is
    a restatement of its published fixed-grid midpoint algorithm
    (FixedGridODESolver.integrate + Midpoint._step_func): the grid is `t`
    itself; per interval  dt = t1 - t0;  f0 = f(t0, y0);
    y1 = y0 + dt * f(t0 + dt/2, y0 + f0 * dt/2);  atol/rtol are ignored; the
    solution at every grid point is stacked.  PARITY UNPINNED: no reference test
    or golden vector pins the sampler.
  * gateloop_transformer.SimpleGateLoopLayer (call sites :31,:399,:466) --
    restated in oracle/restate.py (GateLoopRestated); PARITY UNPINNED.
  * naturalspeech2_pytorch...generate_mask_from_repeats (call site :690, DurationPredictor only) -- restated in
    oracle/restate.py; PARITY UNPINNED.
"""
import os
import sys
import types
import importlib

import torch
from torch import nn

REFERENCE_ROOT = os.environ.get("VBX_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "voicebox_pytorch", "voicebox_pytorch.py"))


def odeint_fixed_grid_midpoint(fn, y0, t, *, atol=None, rtol=None, method="midpoint", **_):
    """Restatement of torchdiffeq.odeint(..., method='midpoint') on the grid t."""
    assert method == "midpoint", "only the fixed-grid midpoint path is restated"
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        half_dt = 0.5 * dt
        f0 = fn(t0, y)
        y_mid = y + f0 * half_dt
        y = y + dt * fn(t0 + half_dt, y_mid)
        ys.append(y)
    return torch.stack(ys, dim=0)


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Placeholder(nn.Module):
    """Stands in for classes the unconditional path never instantiates."""

    def __init__(self, *a, **k):
        super().__init__()


def _identity_decorator(fn=None, **_):
    if fn is None:
        return lambda f: f
    return fn


def install_stubs():
    if "beartype" not in sys.modules:
        bt = _module("beartype", beartype=_identity_decorator)
        import typing

        _module("beartype.typing", Tuple=typing.Tuple, Optional=typing.Optional, List=typing.List,
                Union=typing.Union, Callable=typing.Callable, Dict=typing.Dict)
        _module("beartype.door", is_bearable=lambda *_a, **_k: True)
        bt.typing = sys.modules["beartype.typing"]
        bt.door = sys.modules["beartype.door"]
    if "torchdiffeq" not in sys.modules:
        _module("torchdiffeq", odeint=odeint_fixed_grid_midpoint)
    if "torchode" not in sys.modules:
        _module("torchode", Tsit5=_Placeholder, ODETerm=_Placeholder, IntegralController=_Placeholder,
                AutoDiffAdjoint=_Placeholder, InitialValueProblem=_Placeholder)
    if "naturalspeech2_pytorch" not in sys.modules:
        _module("naturalspeech2_pytorch")
        _module("naturalspeech2_pytorch.aligner", Aligner=_Placeholder, ForwardSumLoss=_Placeholder,
                BinLoss=_Placeholder, maximum_path=lambda *a, **k: None)
        _module("naturalspeech2_pytorch.utils")
        _module("naturalspeech2_pytorch.utils.tokenizer", Tokenizer=_Placeholder)
        from oracle.restate import generate_mask_from_repeats  # restated third-party index logic (DurationPredictor :690)

        _module("naturalspeech2_pytorch.naturalspeech2_pytorch", generate_mask_from_repeats=generate_mask_from_repeats)
    if "audiolm_pytorch" not in sys.modules:
        _module("audiolm_pytorch", EncodecWrapper=_Placeholder, HubertWithKmeans=_Placeholder)
    if "spear_tts_pytorch" not in sys.modules:
        _module("spear_tts_pytorch", TextToSemantic=_Placeholder)
    if "gateloop_transformer" not in sys.modules:
        from oracle.restate import GateLoopRestated  # restated third-party arithmetic

        _module("gateloop_transformer", SimpleGateLoopLayer=GateLoopRestated)
    if "torchaudio" not in sys.modules:
        ta = _module("torchaudio")
        ta.transforms = _module("torchaudio.transforms", Spectrogram=_Placeholder, MelScale=_Placeholder,
                                AmplitudeToDB=_Placeholder)
        ta.functional = _module("torchaudio.functional", DB_to_amplitude=lambda *a, **k: None,
                                resample=lambda *a, **k: None)
        ta.load = lambda *a, **k: None
    if "vocos" not in sys.modules:
        _module("vocos", Vocos=_Placeholder)


_cached = None


def load_reference():
    """Returns the reference's `voicebox_pytorch.voicebox_pytorch` module (unmodified source)."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference sources not present (expected only in the build container)")
    install_stubs()
    import warnings

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # import the model file directly: the package __init__ drags in trainer/data (accelerate etc.)
        pkg = types.ModuleType("voicebox_pytorch")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "voicebox_pytorch")]
        sys.modules.setdefault("voicebox_pytorch", pkg)
        mod = importlib.import_module("voicebox_pytorch.voicebox_pytorch")
    _cached = mod
    return mod
