"""Anti-aliased activation on the MI355X.  Drop-in for both of the reference's implementations:
  * alias_free_activation/torch/act.py:8-30 (Activation1d = UpSample1d -> act -> DownSample1d), and
  * alias_free_activation/cuda/activation1d.py:35-77 + anti_alias_activation_cuda.cu (fused, forward only),
with the same sub-module / buffer names (`act`, `upsample.filter`, `downsample.lowpass.filter`).
forward() is ONE HIP launch (csrc/aa_snake.hip) issued through torch.ops.dmel_hip.aa_snake; when gradients are required it is
differentiable through the native backward kernel (dmel_aa_snake_backward_f32) -- the reference's fused kernel has none
(cuda/activation1d.py:29-32)."""
from __future__ import annotations

import math

import torch
from torch import nn

from ..... import _lib, torch_ops  # noqa: F401  (registers torch.ops.dmel_hip.*)
from ..activations import Snake, SnakeBeta


def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """alias_free_activation/torch/filter.py:30-62 -> (1, 1, kernel_size).  Runs once at construction."""
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
    time = (torch.arange(-half_size, half_size) + 0.5) if even else (torch.arange(kernel_size) - half_size)
    if cutoff == 0:
        return torch.zeros(1, 1, kernel_size)
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.view(1, 1, kernel_size)


class _FilterHolder(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.register_buffer("filter", filt)


class _DownSample(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.lowpass = _FilterHolder(filt)


class Activation1d(nn.Module):
    def __init__(self, activation, up_ratio: int = 2, down_ratio: int = 2, up_kernel_size: int = 12,
                 down_kernel_size: int = 12):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise NotImplementedError("the fused kernel is built for ratio 2 / 12 taps (the only setting BigVGAN uses)")
        if not isinstance(activation, (Snake, SnakeBeta)):
            raise NotImplementedError("activation must be Snake or SnakeBeta")
        self.up_ratio, self.down_ratio = up_ratio, down_ratio
        self.act = activation
        self.upsample = _FilterHolder(kaiser_sinc_filter1d(0.5 / up_ratio, 0.6 / up_ratio, up_kernel_size))
        self.downsample = _DownSample(kaiser_sinc_filter1d(0.5 / down_ratio, 0.6 / down_ratio, down_kernel_size))

    def forward(self, x):
        _lib.require_cuda(x, "x")
        if x.ndim != 3 or x.shape[1] != self.act.in_features:
            raise ValueError(f"expected (B, {self.act.in_features}, T), got {tuple(x.shape)}")
        beta = self.act.beta if isinstance(self.act, SnakeBeta) else None
        # torch.ops.dmel_hip.aa_snake (dmel_codec_amd/torch_ops.py): one launch; differentiable through the native backward kernel.
        # The two filter buffers go in as they are -- the op keeps their 12 host taps cached per (storage, version), so there is no
        # device-to-host copy (and no stream sync) per call; up and down filters may differ, as in fwd_cuda's signature.
        return torch.ops.dmel_hip.aa_snake(x, self.act.alpha, beta, self.upsample.filter, self.downsample.lowpass.filter,
                                           bool(self.act.alpha_logscale))
