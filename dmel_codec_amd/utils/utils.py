"""Mask helpers of the codec path and the checkpoint finder of train_codec.py.  Mirrors dmel_codec/utils/utils.py:11-21,48-67
(reference)."""
from __future__ import annotations

import glob
import os

import torch


def find_lastest_ckpt(directory):
    """utils/utils.py:11-21 (name spelled as there): newest `*.ckpt` of `directory` by modification time, or None."""
    if directory is None:
        return None
    found = glob.glob(os.path.join(directory, "*.ckpt"))
    return max(found, key=os.path.getmtime) if found else None


def sequence_mask(length: torch.Tensor, max_length: int | None = None) -> torch.Tensor:
    """utils/utils.py:48-55: length (B,) or (1, B) -> bool (B, max_length)."""
    if length.ndim == 2:
        length = length.squeeze(0)
    if max_length is None:
        max_length = int(length.max())
    steps = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return steps.unsqueeze(0) < length.unsqueeze(1)


def avg_with_mask(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:58-67"""
    assert mask.dtype == torch.float, "Mask should be float"
    if mask.ndim == 2:
        mask = mask.unsqueeze(1)
    if mask.shape[1] == 1:
        mask = mask.expand_as(x)
    return (x * mask).sum() / mask.sum()
