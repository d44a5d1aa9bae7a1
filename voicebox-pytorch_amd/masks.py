"""Span-mask helpers of the CFM training objective -- integer/boolean host logic, bit-exact with the
reference (voicebox_pytorch.py:68-87, 121-150).  Plain torch ops on whatever device the inputs live on
(IEEE fp32 multiply + truncation give identical results on CPU and GPU); RNG draws can be injected
through `rng_override` so parity tests feed both sides the same numbers.
"""
import contextlib
import threading

import torch

_tls = threading.local()


@contextlib.contextmanager
def rng_override(**draws):
    """Inject the training-step RNG draws (test hook; public signatures stay those of the reference).
    Keys: x0 (:1399), times (:1403), frac_lengths (:1025), rand (:146), y0 (:1289)."""
    prev = getattr(_tls, "draws", None)
    _tls.draws = dict(draws)
    try:
        yield
    finally:
        _tls.draws = prev


def take_draw(name):
    d = getattr(_tls, "draws", None)
    if d is None:
        return None
    return d.get(name)


def prob_mask_like(shape, prob, device):  # voicebox_pytorch.py:68-74
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    if prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def reduce_masks_with_and(*masks):  # voicebox_pytorch.py:76-87
    out = None
    for m in masks:
        if m is None:
            continue
        out = m if out is None else (out & m)
    return out


def mask_from_start_end_indices(seq_len, start, end):  # voicebox_pytorch.py:121-135
    assert start.shape == end.shape
    pos = torch.arange(seq_len, device=start.device, dtype=torch.long)
    pos = pos.reshape(*((1,) * start.ndim), seq_len)
    lo = start.long().unsqueeze(-1)  # float starts truncate toward zero
    hi = end.long().unsqueeze(-1)
    return (pos >= lo) & (pos < hi)


def mask_from_frac_lengths(seq_len, frac_lengths):  # voicebox_pytorch.py:137-150
    lengths = (frac_lengths * seq_len).long()
    max_start = seq_len - lengths
    rand = take_draw("rand")
    if rand is None:
        rand = torch.zeros_like(frac_lengths).float().uniform_(0, 1)
    else:
        rand = rand.to(frac_lengths.device).float()
    start = (max_start * rand).clamp(min=0)
    return mask_from_start_end_indices(seq_len, start, start + lengths)
