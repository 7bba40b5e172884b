"""Sample-rate conversion on the MI355X: `torchaudio.functional.resample(x, orig_freq, new_freq)` as the reference calls it in
LogMelSpectrogram.forward(x, sample_rate=...) (utils/spectrogram.py:122-123; torchaudio's defaults: sinc_interp_hann,
lowpass_filter_width 6, rolloff 0.99).  The polyphase filter bank is built once per (orig, new, device) on the host, the filtering is
one HIP launch (csrc/small_ops.hip: resample_kernel, through torch.ops.dmel_hip.resample)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .. import _lib

_banks: dict = {}


def sinc_resample_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(new_freq // gcd, 2 * width + orig_freq // gcd) float32 filter bank and `width`, torchaudio's `_get_sinc_resample_kernel` for
    resampling_method="sinc_interp_hann", evaluated in float32 like torchaudio does for float32 waveforms."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    if lowpass_filter_width <= 0:
        raise ValueError("Low pass filter width should be positive.")
    base_freq = np.float32(min(orig, new) * rolloff)
    width = int(math.ceil(lowpass_filter_width * orig / (min(orig, new) * rolloff)))
    f32 = np.float32
    idx = (np.arange(-width, width + orig, dtype=np.float32) / f32(orig))[None, :]
    t = (np.arange(0, -new, -1, dtype=np.float32) / f32(new))[:, None] + idx
    t = t * base_freq
    t = np.clip(t, f32(-lowpass_filter_width), f32(lowpass_filter_width))
    window = np.cos(t * f32(math.pi) / f32(lowpass_filter_width) / f32(2)) ** 2
    t = t * f32(math.pi)
    scale = f32(base_freq / f32(orig))
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, f32(1.0), np.sin(t) / t)
    kern = (kern * window * scale).astype(np.float32)
    return kern, width, orig, new


@torch.library.custom_op("dmel_hip::resample", mutates_args=(), device_types="cuda")
def _resample_op(x: torch.Tensor, bank: torch.Tensor, orig: int, new: int, width: int) -> torch.Tensor:
    B, L = x.shape
    Lout = (new * L + orig - 1) // orig
    y = torch.empty(B, Lout, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dmel_resample_f32(x.data_ptr(), y.data_ptr(), bank.data_ptr(), B, L, Lout, orig, new, width,
                                                _lib.stream_ptr()), "resample")
    return y


@_resample_op.register_fake
def _(x, bank, orig, new, width):
    return x.new_empty((x.shape[0], (new * x.shape[1] + orig - 1) // orig), dtype=torch.float32)


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """waveform (..., L) fp32 on the GPU -> (..., ceil(new_freq * L / orig_freq))."""
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Original frequency and desired frequecy should be positive")
    if int(orig_freq) == int(new_freq):
        return waveform
    _lib.require_cuda(waveform, "waveform")
    key = (int(orig_freq), int(new_freq), str(waveform.device))
    if key not in _banks:
        kern, width, orig, new = sinc_resample_bank(orig_freq, new_freq)
        _banks[key] = (torch.from_numpy(kern).to(waveform.device).contiguous(), width, orig, new)
    bank, width, orig, new = _banks[key]
    shape = waveform.shape
    x = waveform.float().reshape(-1, shape[-1]).contiguous()
    y = torch.ops.dmel_hip.resample(x, bank, orig, new, width)
    return y.reshape(*shape[:-1], y.shape[-1])
