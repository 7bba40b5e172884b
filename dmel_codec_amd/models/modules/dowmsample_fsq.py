"""Down/up-sampling finite scalar quantiser on the MI355X.  Drop-in for
dmel_codec/models/modules/dowmsample_fsq.py (reference; file name spelled as there): same class name, ctor kwargs
and state-dict keys, including those of the third-party GroupedResidualFSQ it owns
(`residual_fsq.rvqs.{g}.project_in|project_out.{weight,bias}`).  encode()/decode() run the native path."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch
from torch import nn

from ... import _lib
from ._native import NativeModule
from .firefly import ConvNeXtBlock


@dataclass
class FSQResult:
    z: torch.Tensor
    codes: torch.Tensor
    latents: torch.Tensor


class _ResidualFSQParams(nn.Module):
    """Parameter container with vector_quantize_pytorch.ResidualFSQ's projection names."""

    def __init__(self, dim: int, n_levels: int):
        super().__init__()
        self.project_in = nn.Linear(dim, n_levels)
        self.project_out = nn.Linear(n_levels, dim)


class _GroupedResidualFSQParams(nn.Module):
    """Parameter container with vector_quantize_pytorch.GroupedResidualFSQ's layout (`rvqs` ModuleList)."""

    def __init__(self, dim: int, groups: int, levels):
        super().__init__()
        assert dim % groups == 0
        self.dim, self.groups, self.levels = dim, groups, list(levels)
        self.rvqs = nn.ModuleList([_ResidualFSQParams(dim // groups, len(levels)) for _ in range(groups)])


class _QuantizerTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, z, *params):
        L = _lib.lib()
        G = module.groups
        B, T = z.shape[0] // G, z.shape[2]
        T4 = T // math.prod(module.downsample_factor)
        zq = torch.empty_like(z)
        ids = torch.empty(B, G, T4, dtype=torch.int32, device=z.device)
        lat = torch.empty(z.shape[0], z.shape[1], T4, dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            h = module.native()
            ws = torch.empty(L.dmel_quantizer_train_workspace_bytes(h, B, T), dtype=torch.uint8, device=z.device)
            _lib.check(L.dmel_quantizer_forward_train(h, z.data_ptr(), zq.data_ptr(), ids.data_ptr(), lat.data_ptr(), B, T, ws.data_ptr(),
                                                      ws.numel(), _lib.stream_ptr()), "quantizer_forward_train")
        ctx.module, ctx.handle, ctx.ws = module, h, ws
        module._begin_train_call(ctx)
        ctx.save_for_backward(z)
        ctx.mark_non_differentiable(ids, lat)
        return zq, ids, lat

    @staticmethod
    def backward(ctx, dzq, _dids, _dlat):
        module, h, ws = ctx.module, ctx.handle, ctx.ws
        (z,) = ctx.saved_tensors
        module._check_train_call(ctx)
        L = _lib.lib()
        B, T = z.shape[0] // module.groups, z.shape[2]
        dzq = dzq.float().contiguous()
        dz = torch.empty_like(z)
        with torch.cuda.device(z.device):
            flat = torch.empty(L.dmel_quantizer_grad_floats(h), dtype=torch.float32, device=z.device)
            _lib.check(L.dmel_quantizer_backward(h, z.data_ptr(), dzq.data_ptr(), dz.data_ptr(), flat.data_ptr(), B, T, ws.data_ptr(),
                                                 ws.numel(), _lib.stream_ptr()), "quantizer_backward")
        slots = []
        off, num = C.c_int64(), C.c_int64()
        for key, prm in module.named_parameters():
            _lib.check(L.dmel_quantizer_grad_slot(h, key.encode(), C.byref(off), C.byref(num)), "quantizer_grad_slot")
            slots.append((prm, off.value, num.value))
        grads = module._deliver_grads(flat, slots, ctx.needs_input_grad[2:])
        return (None, dz if ctx.needs_input_grad[1] else None, *grads)


class DownsampleFiniteScalarQuantize(NativeModule):
    """dowmsample_fsq.py:19-147.  fsq_prebound (extension): whether ResidualFSQ.forward bounds its input before the
    first quantiser (newer vector_quantize_pytorch releases do) -- default True."""

    _destroy_symbol = "dmel_quantizer_destroy"
    _set_symbol = "dmel_quantizer_set_tensor"
    _finalize_symbol = "dmel_quantizer_finalize"
    _train_precision_symbol = "dmel_quantizer_set_train_precision"
    _refresh_symbol = "dmel_quantizer_refresh"

    def __init__(self, input_dim: int = 512, n_codebooks: int = 9, n_groups: int = 1, levels=(8, 5, 5, 5),
                 downsample_factor=(2, 2), downsample_dims=None, is_dmel: bool = False, fsq_prebound: bool = True):
        super().__init__()
        if not is_dmel:
            raise NotImplementedError("only the is_dmel=True layout (every reference codec config) is built")
        if n_codebooks != 1:
            raise NotImplementedError("n_codebooks != 1 (residual quantisers) is not used by the codec configs")
        if downsample_dims is None:
            downsample_dims = [input_dim for _ in range(len(downsample_factor))]
        if any(d != input_dim for d in downsample_dims):
            raise NotImplementedError("downsample_dims different from input_dim are not built")
        self.is_dmel, self.groups = is_dmel, n_groups
        self.levels = list(levels)
        self.fsq_prebound = bool(fsq_prebound)
        #: strict encode (extension): project_in + bound in float64 inside encode() -- ids then depend only on the features, not on
        #: a summation order (include/dmel_hip.h: dmel_quantizer_set_strict).  Plain attribute: set it before or after .to(device).
        self.strict_encode = False
        self.input_dim = input_dim
        all_dims = (input_dim // n_groups,) + tuple(d // n_groups for d in downsample_dims)
        self.residual_fsq = _GroupedResidualFSQParams(dim=input_dim, groups=n_groups, levels=levels)
        self.downsample_factor = tuple(downsample_factor)
        self.downsample_dims = downsample_dims
        self.downsample = nn.Sequential(*[
            nn.Sequential(nn.Conv1d(all_dims[idx], all_dims[idx + 1], kernel_size=factor, stride=factor),
                          ConvNeXtBlock(dim=all_dims[idx + 1]))
            for idx, factor in enumerate(downsample_factor)])
        self.upsample = nn.Sequential(*[
            nn.Sequential(nn.ConvTranspose1d(all_dims[idx + 1], all_dims[idx], kernel_size=factor, stride=factor),
                          ConvNeXtBlock(dim=all_dims[idx]))
            for idx, factor in reversed(list(enumerate(downsample_factor)))])
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="leaky_relu")
            nn.init.constant_(m.bias, 0)

    def _create_native(self) -> int:
        h = C.c_void_p()
        lv = (C.c_int * len(self.levels))(*self.levels)
        fs = (C.c_int * len(self.downsample_factor))(*self.downsample_factor)
        _lib.check(_lib.lib().dmel_quantizer_create(C.byref(h), self.input_dim, self.groups, lv, len(self.levels), fs,
                                                    len(self.downsample_factor), int(self.fsq_prebound)),
                   "quantizer_create")
        self._create_native_train(h)
        return h.value

    @torch.no_grad()
    def encode(self, z: torch.Tensor, return_prequant: bool = False, return_latents: bool = False):
        """z (B*G, C, T) -> indices (B, G, T // prod(factors)) int32      (dowmsample_fsq.py:124-133)"""
        _lib.require_cuda(z, "z")
        cg = self.input_dim // self.groups
        if z.ndim != 3 or z.shape[1] != cg or z.shape[0] % self.groups:
            raise ValueError(f"expected (B*{self.groups}, {cg}, T), got {tuple(z.shape)}")
        z = z.float().contiguous()
        B, T = z.shape[0] // self.groups, z.shape[2]
        T4 = T // math.prod(self.downsample_factor)
        if T4 < 1:
            raise ValueError(f"T={T} is shorter than the downsampling factor")
        ids = torch.empty(B, self.groups, T4, dtype=torch.int32, device=z.device)
        pre = torch.empty(self.groups, B, T4, len(self.levels), dtype=torch.float32, device=z.device) if return_prequant else None
        L = _lib.lib()
        with torch.cuda.device(z.device):
            h = self.native()
            _lib.check(L.dmel_quantizer_set_strict(h, int(bool(self.strict_encode))), "quantizer_set_strict")
            ws = self._ws.get(L.dmel_quantizer_workspace_bytes(h, B, T), z.device)
            lat = torch.empty(z.shape[0], cg, T4, dtype=torch.float32, device=z.device) if return_latents else None
            _lib.check(L.dmel_quantizer_encode_ex(h, z.data_ptr(), ids.data_ptr(), _lib.ptr(pre), _lib.ptr(lat), B, T, ws.data_ptr(),
                                                  ws.numel(), _lib.stream_ptr()), "quantizer_encode")
        out = (ids,) + ((pre,) if return_prequant else ()) + ((lat,) if return_latents else ())
        return out if len(out) > 1 else ids

    @torch.no_grad()
    def decode(self, indices: torch.Tensor) -> torch.Tensor:
        """indices (B, G, L) -> z (B, G*C, L * prod(factors))             (dowmsample_fsq.py:135-147)"""
        _lib.require_cuda(indices, "indices")
        if indices.ndim != 3 or indices.shape[1] != self.groups:
            raise ValueError(f"expected (B, {self.groups}, L), got {tuple(indices.shape)}")
        n_codes = math.prod(self.levels)
        ids = indices.to(torch.int32).contiguous()
        B, _, T4 = ids.shape
        Tf = T4 * math.prod(self.downsample_factor)
        z = torch.empty(B, self.input_dim, Tf, dtype=torch.float32, device=ids.device)
        L = _lib.lib()
        del n_codes
        with torch.cuda.device(ids.device):
            h = self.native()
            ws = self._ws.get(L.dmel_quantizer_workspace_bytes(h, B, Tf), ids.device)
            _lib.check(L.dmel_quantizer_decode(h, ids.data_ptr(), z.data_ptr(), B, T4, ws.data_ptr(), ws.numel(),
                                               _lib.stream_ptr()), "quantizer_decode")
        return z

    def _create_native_train(self, h):
        if getattr(self, "_want_train", False):
            _lib.check(_lib.lib().dmel_quantizer_enable_training(h, 1), "quantizer_enable_training")

    def forward(self, z) -> FSQResult:
        """Training-time forward (dowmsample_fsq.py:86-122): z (B*G, C, T) -> FSQResult(z=(B, G*C, T), codes=(B, G, T4),
        latents=(B, G*C, T4)); differentiable (straight-through FSQ) through the native backward."""
        _lib.require_cuda(z, "z")
        cg = self.input_dim // self.groups
        if z.ndim != 3 or z.shape[1] != cg or z.shape[0] % self.groups:
            raise ValueError(f"expected (B*{self.groups}, {cg}, T), got {tuple(z.shape)}")
        if z.shape[2] // math.prod(self.downsample_factor) < 1:
            raise ValueError(f"T={z.shape[2]} is shorter than the downsampling factor")
        if not getattr(self, "_want_train", False):
            self._want_train = True
            self._free_native()
        zq, ids, lat = _QuantizerTrainFn.apply(self, z.float().contiguous(), *list(self.parameters()))
        B = z.shape[0] // self.groups
        return FSQResult(z=zq.view(B, self.input_dim, -1), codes=ids, latents=lat.view(B, self.input_dim, -1))
