"""Log-mel front end on the MI355X.  Drop-in for dmel_codec/utils/spectrogram.py (reference):
same class names, ctor kwargs, attributes and forward signature; the six ATen launches of
LinearSpectrogram.forward (spectrogram.py:58-79) are one HIP kernel (csrc/stft_logmel.hip)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor, nn

from .. import _lib, torch_ops  # noqa: F401  (registers torch.ops.dmel_hip.*)


class LinearSpectrogram(nn.Module):
    """Reference: utils/spectrogram.py:7-81.  forward(y) returns the LOG-MEL spectrogram, as the reference's does."""

    def __init__(self, n_fft=2048, win_length=2048, hop_length=512, center=False, num_mels=128, f_min=0,
                 f_max=None, sample_rate=44100, mode="reflect"):
        super().__init__()
        if center:
            raise NotImplementedError("center=True is not used by any reference config and is not built")
        if mode != "reflect":
            raise NotImplementedError(f"pad mode {mode!r}: only 'reflect' (the reference's only setting) is built")
        self.n_fft, self.win_length, self.hop_length = n_fft, win_length, hop_length
        self.center, self.mode = center, mode
        self.f_min, self.f_max = f_min, f_max
        self.num_mels, self.sample_rate = num_mels, sample_rate
        self._plan: Optional[int] = None     # plan handle; cached like mel_basis_cache / hann_window_cache (:43-56)

    def _get_plan(self) -> int:
        if self._plan is None:
            L = _lib.lib()
            h = C.c_void_p()
            window = torch.hann_window(self.win_length, dtype=torch.float32)   # spectrogram.py:53
            _lib.check(L.dmel_stft_plan_create(C.byref(h), self.sample_rate, self.n_fft, self.win_length, self.hop_length,
                                               self.num_mels, float(self.f_min or 0.0),
                                               float(self.f_max) if self.f_max else 0.0, window.data_ptr()),
                       "stft_plan_create")
            self._plan = h.value
        return self._plan

    def mel_basis(self) -> Tensor:
        """Dense (num_mels, n_fft//2+1) Slaney filterbank the plan was built from (host tensor)."""
        out = torch.empty(self.num_mels, self.n_fft // 2 + 1, dtype=torch.float32)
        _lib.check(_lib.lib().dmel_stft_plan_mel_basis(self._get_plan(), out.data_ptr()), "stft_plan_mel_basis")
        return out

    def __getstate__(self):      # copies / pickles create their own plan
        st = self.__dict__.copy()
        st["_plan"] = None
        return st

    def __del__(self):
        if getattr(self, "_plan", None) is not None and _lib._lib is not None:
            _lib._lib.dmel_stft_plan_destroy(self._plan)
            self._plan = None

    def forward(self, y: Tensor, lengths: Optional[Tensor] = None) -> Tensor:
        """y: (B, L) or (B, 1, L) fp32 on the GPU -> (B, num_mels, L // hop).
        lengths (extension, optional): (B,) int64 sample counts; frames >= lengths // hop are zeroed, which is the
        `mels * mask` of codec_lit_modules.py:492-506 fused into the store."""
        _lib.require_cuda(y, "audio")
        if y.ndim == 3:
            if y.shape[1] != 1:
                raise ValueError(f"expected mono audio (B, 1, L), got {tuple(y.shape)}")
            y = y[:, 0]
        if y.ndim != 2:
            raise ValueError(f"expected (B, L) or (B, 1, L), got {tuple(y.shape)}")
        lens = None
        if lengths is not None:
            lens = lengths.reshape(-1).to(device=y.device, dtype=torch.int64).contiguous()
            if lens.numel() != y.shape[0]:
                raise ValueError("lengths must have one entry per batch item")
        # through PyTorch's dispatcher (dmel_codec_amd/torch_ops.py): torch.ops.dmel_hip.stft_logmel -> dmel_stft_logmel_f32; the plan
        # (window, twiddles, sparse mel bands) is cached there per (configuration, device), as the reference caches its mel basis and
        # window per device (spectrogram.py:43-56)
        return torch.ops.dmel_hip.stft_logmel(y, lens, self.sample_rate, self.n_fft, self.win_length, self.hop_length, self.num_mels,
                                              float(self.f_min or 0.0), float(self.f_max) if self.f_max else 0.0)


class LogMelSpectrogram(nn.Module):
    """Reference: utils/spectrogram.py:84-127."""

    def __init__(self, sample_rate=44100, n_fft=2048, win_length=2048, hop_length=512, n_mels=128, center=False,
                 f_min=0.0, f_max=None):
        super().__init__()
        self.sample_rate, self.n_fft, self.win_length, self.hop_length = sample_rate, n_fft, win_length, hop_length
        self.center, self.n_mels, self.f_min = center, n_mels, f_min
        self.f_max = f_max or float(sample_rate // 2)
        self.spectrogram = LinearSpectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop_length, center=center,
                                             num_mels=n_mels, f_min=f_min, f_max=f_max, sample_rate=sample_rate,
                                             mode="reflect")

    def forward(self, x: Tensor, return_linear: bool = False, sample_rate: int = None,
                lengths: Optional[Tensor] = None) -> Tensor:
        if sample_rate is not None and sample_rate != self.sample_rate:
            # spectrogram.py:122-123: torchaudio.functional.resample(x, orig_freq=sample_rate, new_freq=self.sample_rate)
            from .resample import resample
            x = resample(x, orig_freq=sample_rate, new_freq=self.sample_rate)
            if lengths is not None:
                lengths = (lengths.to(torch.int64) * self.sample_rate + sample_rate - 1) // sample_rate
        return self.spectrogram(x, lengths=lengths)
