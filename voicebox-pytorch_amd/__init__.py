"""voicebox-pytorch_amd: MI355X (gfx950) native hot path of lucidrains/voicebox-pytorch.

Public API mirrors the reference package (voicebox_pytorch/__init__.py:1-15) for the hot path:
VoiceBox, ConditionalFlowMatcherWrapper, Transformer, Attend, DurationPredictor (inference), VoiceBoxTrainer (latents).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
try:  # model classes need torch; keep `_lib` importable on its own
    from .masks import mask_from_frac_lengths, mask_from_start_end_indices, prob_mask_like, reduce_masks_with_and  # noqa: F401
    from .model import VoiceBox, ConditionalFlowMatcherWrapper, Transformer, Attend  # noqa: F401
    from .trainer import VoiceBoxTrainer  # noqa: F401
    from .duration import DurationPredictor  # noqa: F401
    from .engine import precise_mode, set_precise, precise_enabled  # noqa: F401

    __all__ += ["VoiceBox", "ConditionalFlowMatcherWrapper", "Transformer", "Attend", "VoiceBoxTrainer", "DurationPredictor", "mask_from_frac_lengths",
                "mask_from_start_end_indices", "prob_mask_like", "reduce_masks_with_and", "precise_mode", "set_precise", "precise_enabled"]
except ModuleNotFoundError as _e:  # pragma: no cover - only while the package is being bootstrapped
    if "masks" not in str(_e) and "model" not in str(_e):
        raise
