"""Mel-image discriminator on the MI355X.  Mirrors dmel_codec/models/modules/discriminator.py:6-35 (reference): six weight-normed
Conv2d with SiLU in between, state-dict keys `blocks.{i}.bias`, `blocks.{i}.parametrizations.weight.original0|1`.  forward() runs
the native handle (csrc/modules.hip: every (3, kw) Conv2d as three implicit-GEMM launches over padded image rows).  Inference only so
far: its backward -- the last missing piece of training_step (codec_lit_modules.py:214-267) -- is not built, and a call that would
need gradients raises instead of silently detaching."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn
from torch.nn.utils.parametrizations import weight_norm

from ... import _lib
from ._native import NativeModule


class Discriminator(NativeModule):
    _destroy_symbol = "dmel_discriminator_destroy"
    _set_symbol = "dmel_discriminator_set_tensor"
    _finalize_symbol = "dmel_discriminator_finalize"

    def __init__(self):
        super().__init__()
        convs = [(1, 64, (3, 9), 1, (1, 4)), (64, 128, (3, 9), (1, 2), (1, 4)), (128, 256, (3, 9), (1, 2), (1, 4)),
                 (256, 512, (3, 9), (1, 2), (1, 4)), (512, 1024, (3, 3), 1, (1, 1)), (1024, 1, (3, 3), 1, (1, 1))]
        blocks = []
        for idx, (cin, cout, k, stride, pad) in enumerate(convs):
            blocks.append(weight_norm(nn.Conv2d(cin, cout, k, stride, pad)))
            if idx != len(convs) - 1:
                blocks.append(nn.SiLU(inplace=True))
        self.blocks = nn.Sequential(*blocks)

    def _create_native(self) -> int:
        h = C.c_void_p()
        _lib.check(_lib.lib().dmel_discriminator_create(C.byref(h)), "discriminator_create")
        return h.value

    def forward(self, x):
        """x (B, n_mels, T) -> logits (B, n_mels, T')        (discriminator.py:34-35)"""
        _lib.require_cuda(x, "x")
        if x.ndim != 3:
            raise ValueError(f"expected (B, n_mels, T), got {tuple(x.shape)}")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("the discriminator's backward is not built yet (SURVEY.md 8f rank 1); call it under "
                                      "torch.no_grad() for logits only")
        x = x.float().contiguous()
        B, H, W = x.shape
        L = _lib.lib()
        with torch.no_grad(), torch.cuda.device(x.device):
            h = self.native()
            y = torch.empty(B, H, L.dmel_discriminator_out_frames(h, W), dtype=torch.float32, device=x.device)
            ws = self._ws.get(L.dmel_discriminator_workspace_bytes(h, B, H, W), x.device)
            _lib.check(L.dmel_discriminator_forward(h, x.data_ptr(), y.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "discriminator_forward")
        return y
