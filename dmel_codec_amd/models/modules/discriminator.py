"""Mel-image discriminator on the MI355X.  Mirrors dmel_codec/models/modules/discriminator.py:6-35 (reference): six weight-normed
Conv2d with SiLU in between, state-dict keys `blocks.{i}.bias`, `blocks.{i}.parametrizations.weight.original0|1`.  forward() runs
the native handle (csrc/modules.hip: every (3, kw) Conv2d as three implicit-GEMM launches over padded image rows) and is differentiable
through the hand-written backward (dmel_discriminator_forward_train / _backward: transposed convs per kernel row and output phase, 27
single-tap weight-gradient GEMMs per layer, the chain through weight norm)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn
from torch.nn.utils.parametrizations import weight_norm

from ... import _lib
from ._native import NativeModule


class _DiscTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        L = _lib.lib()
        B, H, W = x.shape
        with torch.cuda.device(x.device):
            h = module.native()
            y = torch.empty(B, H, L.dmel_discriminator_out_frames(h, W), dtype=torch.float32, device=x.device)
            ws = torch.empty(L.dmel_discriminator_train_workspace_bytes(h, B, H, W), dtype=torch.uint8, device=x.device)
            _lib.check(L.dmel_discriminator_forward_train(h, x.data_ptr(), y.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(),
                                                          _lib.stream_ptr()), "discriminator_forward_train")
        ctx.module, ctx.handle, ctx.ws, ctx.shape = module, h, ws, (B, H, W)
        module._begin_train_call(ctx)
        return y

    @staticmethod
    def backward(ctx, dy):
        module, h, ws = ctx.module, ctx.handle, ctx.ws
        module._check_train_call(ctx)
        L = _lib.lib()
        B, H, W = ctx.shape
        dy = dy.float().contiguous()
        dx = torch.empty(B, H, W, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dy.device):
            flat = torch.empty(L.dmel_discriminator_grad_floats(h), dtype=torch.float32, device=dy.device)
            _lib.check(L.dmel_discriminator_backward(h, dy.data_ptr(), _lib.ptr(dx), flat.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(),
                                                     _lib.stream_ptr()), "discriminator_backward")
        slots = []
        off, num = C.c_int64(), C.c_int64()
        for key, prm in module.named_parameters():
            _lib.check(L.dmel_discriminator_grad_slot(h, key.encode(), C.byref(off), C.byref(num)), "discriminator_grad_slot")
            slots.append((prm, off.value, num.value))
        grads = module._deliver_grads(flat, slots, ctx.needs_input_grad[2:])
        return (None, dx, *grads)


class Discriminator(NativeModule):
    _destroy_symbol = "dmel_discriminator_destroy"
    _set_symbol = "dmel_discriminator_set_tensor"
    _finalize_symbol = "dmel_discriminator_finalize"
    _train_precision_symbol = "dmel_discriminator_set_train_precision"
    _refresh_symbol = "dmel_discriminator_refresh"

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
        if getattr(self, "_want_train", False):
            _lib.check(_lib.lib().dmel_discriminator_enable_training(h, 1), "discriminator_enable_training")
        return h.value

    def forward(self, x):
        """x (B, n_mels, T) -> logits (B, n_mels, T')        (discriminator.py:34-35)"""
        _lib.require_cuda(x, "x")
        if x.ndim != 3:
            raise ValueError(f"expected (B, n_mels, T), got {tuple(x.shape)}")
        x = x.float().contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if not getattr(self, "_want_train", False):
                self._want_train = True
                self._free_native()
            return _DiscTrainFn.apply(self, x, *list(self.parameters()))
        B, H, W = x.shape
        L = _lib.lib()
        with torch.no_grad(), torch.cuda.device(x.device):
            h = self.native()
            y = torch.empty(B, H, L.dmel_discriminator_out_frames(h, W), dtype=torch.float32, device=x.device)
            ws = self._ws.get(L.dmel_discriminator_workspace_bytes(h, B, H, W), x.device)
            _lib.check(L.dmel_discriminator_forward(h, x.data_ptr(), y.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "discriminator_forward")
        return y
