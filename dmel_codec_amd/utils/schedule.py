"""Warm-up + cosine learning-rate multiplier named by the codec configs
(`lr_lambda._target_: dmel_codec.utils.schedule.get_cosine_schedule_with_warmup_lr_lambda`).

Behaviour follows the reference's utils/schedule.py:4-25: linear ramp 0 -> 1 over the warm-up steps (a warm-up given as
a fraction in (0, 1) is a share of the training steps), then `0.5 * (1 + cos(2 pi cycles p))` over the remaining
progress p in [0, 1], never below `final_lr_ratio`.  Pinned by tests/golden/schedule.npz."""
from __future__ import annotations

from math import cos, pi

__all__ = ["get_cosine_schedule_with_warmup_lr_lambda"]


def _warmup_steps(spec, total: int) -> int | float:
    return int(spec * total) if 0 < spec < 1 else spec


def get_cosine_schedule_with_warmup_lr_lambda(current_step: int, *, num_warmup_steps, num_training_steps: int,
                                              num_cycles: float = 0.5, final_lr_ratio: float = 0.0) -> float:
    warm = _warmup_steps(num_warmup_steps, num_training_steps)
    if current_step < warm:                                   # ramp
        return current_step / float(warm if warm > 1 else 1)
    span = num_training_steps - warm
    p = (current_step - warm) / float(span if span > 1 else 1)
    cosine = (1.0 + cos(2.0 * pi * float(num_cycles) * p)) / 2.0
    return cosine if cosine > final_lr_ratio else final_lr_ratio
