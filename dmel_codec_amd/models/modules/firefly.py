"""ConvNeXtBlock / LayerNorm.  Mirrors dmel_codec/models/modules/firefly.py:306-402 (reference); only the two classes the
codec path uses are provided.  Inside the quantiser the blocks' arithmetic runs in the quantiser's native calls; called on its
own, ConvNeXtBlock.forward runs the standalone native handle (dmel_convnext_*: csrc/small_ops.hip dwconv_ln + two implicit-GEMM
launches) and is differentiable through the hand-written backward (dmel_convnext_forward_train / dmel_convnext_backward)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from ... import _lib
from ._native import NativeModule


class LayerNorm(nn.Module):
    """firefly.py:306-333 (channels_last form; parameter container)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format != "channels_last":
            raise NotImplementedError("only the channels_last form is on the codec path")
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)


class _ConvNeXtTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        L = _lib.lib()
        N, _, T = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            h = module.native()
            ws = torch.empty(L.dmel_convnext_train_workspace_bytes(h, N, T), dtype=torch.uint8, device=x.device)
            _lib.check(L.dmel_convnext_forward_train(h, x.data_ptr(), y.data_ptr(), N, T, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "convnext_forward_train")
        ctx.module, ctx.handle, ctx.ws = module, h, ws
        module._begin_train_call(ctx)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        module, h, ws = ctx.module, ctx.handle, ctx.ws
        (x,) = ctx.saved_tensors
        module._check_train_call(ctx)
        L = _lib.lib()
        N, _, T = x.shape
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            flat = torch.empty(L.dmel_convnext_grad_floats(h), dtype=torch.float32, device=x.device)
            _lib.check(L.dmel_convnext_backward(h, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), flat.data_ptr(), N, T, ws.data_ptr(),
                                                ws.numel(), _lib.stream_ptr()), "convnext_backward")
        slots = []
        off, num = C.c_int64(), C.c_int64()
        for key, prm in module.named_parameters():
            _lib.check(L.dmel_convnext_grad_slot(h, key.encode(), C.byref(off), C.byref(num)), "convnext_grad_slot")
            slots.append((prm, off.value, num.value))
        grads = module._deliver_grads(flat, slots, ctx.needs_input_grad[2:])
        return (None, dx if ctx.needs_input_grad[1] else None, *grads)


class ConvNeXtBlock(NativeModule):
    """firefly.py:337-402."""

    _destroy_symbol = "dmel_convnext_destroy"
    _set_symbol = "dmel_convnext_set_tensor"
    _finalize_symbol = "dmel_convnext_finalize"
    _train_precision_symbol = "dmel_convnext_set_train_precision"

    def __init__(self, dim: int, drop_path: float = 0.0, layer_scale_init_value: float = 1e-6, mlp_ratio: float = 4.0,
                 kernel_size: int = 7, dilation: int = 1):
        super().__init__()
        if kernel_size != 7 or dilation != 1 or mlp_ratio != 4.0 or drop_path != 0.0 or layer_scale_init_value <= 0:
            raise NotImplementedError("only the default ConvNeXtBlock (k=7, mlp 4x, layer scale) is on the codec path")
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=kernel_size, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, int(mlp_ratio * dim))
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(int(mlp_ratio * dim), dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True)
        self.dim = dim

    def _create_native(self) -> int:
        h = C.c_void_p()
        _lib.check(_lib.lib().dmel_convnext_create(C.byref(h), self.dim), "convnext_create")
        if getattr(self, "_want_train", False):
            _lib.check(_lib.lib().dmel_convnext_enable_training(h, 1), "convnext_enable_training")
        return h.value

    def forward(self, x, apply_residual: bool = True):
        """x (N, dim, T) -> (N, dim, T)        (firefly.py:383-402)"""
        if not apply_residual:
            raise NotImplementedError("apply_residual=False is never used on the codec path")
        _lib.require_cuda(x, "x")
        if x.ndim != 3 or x.shape[1] != self.dim:
            raise ValueError(f"expected (N, {self.dim}, T), got {tuple(x.shape)}")
        x = x.float().contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if not getattr(self, "_want_train", False):
                self._want_train = True
                self._free_native()
            return _ConvNeXtTrainFn.apply(self, x, *list(self.parameters()))
        N, _, T = x.shape
        L = _lib.lib()
        y = torch.empty_like(x)
        with torch.no_grad(), torch.cuda.device(x.device):
            h = self.native()
            ws = self._ws.get(L.dmel_convnext_workspace_bytes(h, N, T), x.device)
            _lib.check(L.dmel_convnext_forward(h, x.data_ptr(), y.data_ptr(), N, T, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "convnext_forward")
        return y
