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

    def __init__(self):
        super().__init__()
        self._handle: Optional[int] = None
        self._handle_versions = None
        self._ws = _lib.Workspace()

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
            except Exception:
                getattr(L, self._destroy_symbol)(h)
                raise
            self._handle, self._handle_versions = h, ver
        return self._handle

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
