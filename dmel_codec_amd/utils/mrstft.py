"""Multi-resolution STFT loss on the MI355X.  BASELINE.json's north_star lists it next to the BigVGAN decode side; the reference has
NO such loss (its generator loss is band-weighted mel L1 + LSGAN, codec_lit_modules.py:246-274) -- so there is nothing in the
reference to be on par with: PARITY UNPINNED BY THE REFERENCE.  The definition is the standard one (Parallel WaveGAN): per resolution
    spectral convergence  || |S(y)| - |S(x)| ||_F / || |S(y)| ||_F     +     mean | log|S(y)| - log|S(x)| |
averaged over the resolutions, with the reference's own STFT framing (utils/spectrogram.py:58-76) at each resolution, and it is
checked against a torch.stft restatement in oracle/ref_cpu.py.  The magnitudes come from the fused STFT kernel
(torch.ops.dmel_hip.stft_magnitude, one launch per signal and resolution, mel stage skipped).

Differentiable with respect to `pred` (round 3): torch.ops.dmel_hip.stft_magnitude has a native backward (dmel_stft_magnitude_backward_f32:
the windowed DFT and its transpose as GEMMs on the library's convolution kernel, overlap-add with the reflect padding folded back), so
the loss can train whatever produces the waveform.  In the reference's own training_step nothing does: the vocoder between the trained
networks and a waveform is frozen (codec_lit_modules.py:68-72), so there the loss remains a validation metric; it becomes a training
loss the moment a waveform-producing module is trainable (vocoder fine-tuning, which this package does not build)."""
from __future__ import annotations

from typing import Sequence

import torch
from torch import nn

from .. import torch_ops  # noqa: F401  (registers torch.ops.dmel_hip.*)


class MultiResolutionSTFTLoss(nn.Module):
    def __init__(self, fft_sizes: Sequence[int] = (1024, 2048, 512), hop_sizes: Sequence[int] = (120, 240, 50),
                 win_lengths: Sequence[int] = (600, 1200, 240)):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.resolutions = list(zip(fft_sizes, hop_sizes, win_lengths))

    def forward(self, pred: torch.Tensor, target: torch.Tensor):
        """pred / target: (B, L) or (B, 1, L) on the GPU -> (spectral convergence, log-magnitude L1), each averaged over resolutions.
        Gradients flow to `pred` (the target's magnitudes are constants)."""
        if pred.ndim == 3:
            pred, target = pred[:, 0], target[:, 0]
        sc_total, mag_total = 0.0, 0.0
        for n_fft, hop, win in self.resolutions:
            sp = torch.ops.dmel_hip.stft_magnitude(pred, n_fft, win, hop)
            with torch.no_grad():
                st = torch.ops.dmel_hip.stft_magnitude(target, n_fft, win, hop)
            sc_total = sc_total + torch.linalg.norm(st - sp) / torch.linalg.norm(st)
            mag_total = mag_total + (st.log() - sp.log()).abs().mean()
        n = len(self.resolutions)
        return sc_total / n, mag_total / n
