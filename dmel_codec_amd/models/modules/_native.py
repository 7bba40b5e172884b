"""Shared plumbing of the mirror modules: a lazily built native handle that is rebuilt whenever the module's
parameters change (load_state_dict, in-place edits) and freed with the module."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import nn

from ... import _lib


class NativeModule(nn.Module):
    """Base class: subclasses implement _create_native() -> handle and name their destroy/set/finalize symbols."""

    _destroy_symbol = ""
    _set_symbol = ""
    _finalize_symbol = ""
    _precision_symbol = ""   # modules whose convolutions have the opt-in bf16 mode name their dmel_*_set_precision here
    _refresh_symbol = ""     # modules that can re-pack their weight images from device tensors name their dmel_*_refresh here

    def __init__(self):
        super().__init__()
        self._handle: Optional[int] = None
        self._handle_versions = None
        self._ws = _lib.Workspace()
        self._precision = 0

    # -- handle life cycle ---------------------------------------------------------------------
    def _native_state(self) -> dict:
        return self.state_dict()

    def _versions(self):
        return tuple((id(t), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _create_native(self) -> int:
        raise NotImplementedError

    def _native_state_refs(self):
        """(key, live tensor) of the tensors the native handle is built from -- no copies (device-side refresh)."""
        keys = set(self._native_state().keys())
        return [(k, p.data) for k, p in self.named_parameters() if k in keys]

    def native(self) -> int:
        ver = self._versions()
        if self._refresh_symbol and self._handle is not None and ver != self._handle_versions:
            # only the VALUES of CUDA parameters changed (an optimiser step): re-pack the existing handle's weight images on the
            # device instead of rebuilding the handle through the host
            old = self._handle_versions
            same = old is not None and len(old) == len(ver) and all(a[0] == b[0] for a, b in zip(old, ver))
            items = self._native_state_refs()
            if same and items and all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() for _, v in items):
                keys = (C.c_char_p * len(items))(*[k.encode() for k, _ in items])
                ptrs = (C.c_void_p * len(items))(*[v.data_ptr() for _, v in items])
                with torch.cuda.device(items[0][1].device):
                    _lib.check(getattr(_lib.lib(), self._refresh_symbol)(self._handle, len(items), keys, ptrs, _lib.stream_ptr()),
                               f"{type(self).__name__}.refresh")
                self._handle_versions = ver
                return self._handle
        if self._handle is None or ver != self._handle_versions:
            self._free_native()
            L = _lib.lib()
            h = self._create_native()
            try:
                _lib.set_tensors(getattr(L, self._set_symbol), h, self._native_state(), type(self).__name__)
                _lib.check(getattr(L, self._finalize_symbol)(h), f"{type(self).__name__}.finalize")
                if self._precision:
                    _lib.check(getattr(L, self._precision_symbol)(h, self._precision), f"{type(self).__name__}.set_precision")
            except Exception:
                getattr(L, self._destroy_symbol)(h)
                raise
            self._handle, self._handle_versions = h, ver
        return self._handle

    def set_precision(self, precision) -> None:
        """"fp32" (default; the parity path), "fp32_mfma" (force the native fp32 MFMA kernel) or "bf16": convolution operands
        rounded to bf16, fp32 accumulation, fp32 tensors everywhere else (include/dmel_hip.h, DMEL_PRECISION_*).  Accepts the
        strings, torch dtypes or the DMEL_PRECISION_* integers."""
        table = {"fp32": 0, "float32": 0, torch.float32: 0, 0: 0, "bf16": 1, "bfloat16": 1, torch.bfloat16: 1, 1: 1,
                 "fp32_mfma": 2, 2: 2}
        if precision not in table:
            raise ValueError(f"precision must be 'fp32', 'fp32_mfma' or 'bf16', got {precision!r}")
        if table[precision] and not self._precision_symbol:
            raise NotImplementedError(f"{type(self).__name__} has no bf16 mode")
        self._precision = table[precision]
        if self._handle is not None and self._precision_symbol:
            _lib.check(getattr(_lib.lib(), self._precision_symbol)(self._handle, self._precision),
                       f"{type(self).__name__}.set_precision")

    def _free_native(self):
        if getattr(self, "_handle", None) is not None and _lib._lib is not None:
            getattr(_lib._lib, self._destroy_symbol)(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._free_native()
        except Exception:
            pass

    def _device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")
