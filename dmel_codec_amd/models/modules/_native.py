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

    def native(self) -> int:
        ver = self._versions()
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
