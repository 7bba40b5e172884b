"""Shared plumbing of the mirror modules: a lazily built native handle that is rebuilt whenever the module's
parameters change (load_state_dict, in-place edits) and freed with the module."""
from __future__ import annotations

import ctypes as C
import weakref
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
    _train_precision_symbol = ""   # trainable modules name their dmel_*_set_train_precision here (bf16 training mode)

    def __init__(self):
        super().__init__()
        self._handle: Optional[int] = None
        self._handle_versions = None
        self._ws = _lib.Workspace()
        self._precision = 0
        self._train_precision = 0
        self._generation = 0          # bumped whenever the native weight images change (rebuild or device-side re-pack)
        self._grad_sink = None        # armed by ddp.GradReducer for the duration of one backward pass
        # autograd contexts of training forwards whose backward has not run yet.  WEAK references: a graph that is dropped without a
        # backward (a grad-enabled forward in a callback, a step that raised) leaves the set by itself instead of blocking every later
        # gradient exchange of this module
        self._train_calls = weakref.WeakSet()
        self._backward_ctx = None     # the context whose backward is running (set by _check_train_call)
        self._grads_delivered = False  # while armed: parameter gradients were pointed at the flat buffer ...
        self._grads_submitted = False  # ... and the buffer went to the reducer (GradReducer.finish checks the pair)

    # -- copies --------------------------------------------------------------------------------
    def __getstate__(self):
        """copy.deepcopy / pickle: a copy builds its OWN native handle and workspace on first use (the handle is an address inside the
        library: copied by value, the copy would free the original's, or run in its scratch)."""
        st = self.__dict__.copy()
        st["_handle"], st["_handle_versions"] = None, None
        st["_ws"] = _lib.Workspace()
        st.pop("_train_calls", None)          # a WeakSet does not pickle; __setstate__ makes a new one
        st["_grad_sink"], st["_backward_ctx"] = None, None
        st["_grads_delivered"], st["_grads_submitted"] = False, False
        return st

    def __setstate__(self, state):
        super().__setstate__(state)
        self._train_calls = weakref.WeakSet()

    # -- handle life cycle ---------------------------------------------------------------------
    def _native_state(self) -> dict:
        return self.state_dict()

    def _versions(self):
        # the device is part of the key: module.to(other_device) keeps ids and versions but the handle's buffers live on the old one
        return tuple((id(t), t._version, str(t.device)) for t in list(self.parameters()) + list(self.buffers()))

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
            same = old is not None and len(old) == len(ver) and all(a[0] == b[0] and a[2] == b[2] for a, b in zip(old, ver))
            items = self._native_state_refs()
            if same and items and all(v.is_cuda and v.dtype == torch.float32 and v.is_contiguous() for _, v in items):
                keys = (C.c_char_p * len(items))(*[k.encode() for k, _ in items])
                ptrs = (C.c_void_p * len(items))(*[v.data_ptr() for _, v in items])
                with torch.cuda.device(items[0][1].device):
                    _lib.check(getattr(_lib.lib(), self._refresh_symbol)(self._handle, len(items), keys, ptrs, _lib.stream_ptr()),
                               f"{type(self).__name__}.refresh")
                self._handle_versions = ver
                self._generation += 1
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
                if self._train_precision:
                    _lib.check(getattr(L, self._train_precision_symbol)(h, self._train_precision),
                               f"{type(self).__name__}.set_train_precision")
            except Exception:
                getattr(L, self._destroy_symbol)(h)
                raise
            self._handle, self._handle_versions = h, ver
            self._generation += 1
        return self._handle

    # -- training calls: shared bookkeeping of the autograd Functions ---------------------------
    @property
    def _pending_train(self) -> int:
        """Training forwards of this module whose graph is still alive and whose backward has not run."""
        return len(self._train_calls)

    def _begin_train_call(self, ctx) -> None:
        """Called by a training Function's forward after module.native(): remembers which weight images the saved activations
        belong to.  An optimiser step between this forward and its backward re-packs the images in place (same handle address), which
        torch autograd would report as an in-place modification; the generation counter makes the native backward report it too."""
        ctx.generation = self._generation
        self._train_calls.add(ctx)

    def _check_train_call(self, ctx) -> None:
        self._backward_ctx = ctx
        if self._generation != ctx.generation or self._handle is None:
            self._train_calls.discard(ctx)
            self._backward_ctx = None
            raise RuntimeError(f"{type(self).__name__}: parameters changed (optimiser step / load_state_dict / .to()) between a "
                               "training forward and its backward")

    def _deliver_grads(self, flat: torch.Tensor, slots, needs, streamed: bool = False):
        """Hand the flat native gradient buffer to autograd -- or, while a ddp.GradReducer is armed on this module, straight to the
        parameters and the reducer.  slots: [(parameter, offset, numel)] in the order of the Function's parameter inputs.
        Unarmed: returns one view of `flat` per parameter (autograd accumulates them as usual).
        Armed: parameter gradients become VIEWS of `flat` (earlier gradients are added into it first), `flat` is submitted to the
        reducer once the last outstanding backward of this module has run, and autograd receives None for the parameters -- the
        all-reduce works in place on the buffer the optimiser will read, no flatten / copy-back passes.  streamed=True: the C side
        already submitted every region of `flat` through the on_ready hook."""
        if self._backward_ctx is not None:
            self._train_calls.discard(self._backward_ctx)
            self._backward_ctx = None
        views = [flat[o:o + n].view(p.shape) if need else None for (p, o, n), need in zip(slots, needs)]
        sink = self._grad_sink
        if sink is None:
            return views
        for (p, _, _), v in zip(slots, views):
            if v is None:
                continue
            if p.grad is not None:
                v.add_(p.grad)
            p.grad = v
        self._grads_delivered = True
        if streamed:
            self._grads_submitted = True
        elif self._pending_train == 0:
            sink.submit(flat)
            self._grads_submitted = True
        return [None] * len(views)

    def _can_stream_grads(self, params, needs) -> bool:
        """Per-block submission from inside the native backward is possible when this is the module's only outstanding backward and
        no earlier gradient has to be added into the buffer first."""
        return (self._grad_sink is not None and self._pending_train == 1 and all(needs)
                and all(p.grad is None for p in params))

    def set_precision(self, precision) -> None:
        """"fp32" (default; the parity path: fp32-grade products, the library picks the construction), "fp32_f16x2" (force the
        three-product fp16 split), "fp32_bf16x3" (force the six-product bf16 split), "fp32_mfma" (force the native fp32 MFMA kernel) or
        "bf16": convolution operands rounded to bf16, fp32 accumulation, fp32 tensors everywhere else (include/dmel_hip.h,
        DMEL_PRECISION_*).  Accepts the strings, torch dtypes or the DMEL_PRECISION_* integers."""
        table = {"fp32": 0, "float32": 0, torch.float32: 0, 0: 0, "bf16": 1, "bfloat16": 1, torch.bfloat16: 1, 1: 1,
                 "fp32_mfma": 2, 2: 2, "fp32_f16x2": 3, 3: 3, "fp32_bf16x3": 4, 4: 4}
        if precision not in table:
            raise ValueError(f"precision must be 'fp32', 'fp32_f16x2', 'fp32_bf16x3', 'fp32_mfma' or 'bf16', got {precision!r}")
        if table[precision] and not self._precision_symbol:
            raise NotImplementedError(f"{type(self).__name__} has no bf16 mode")
        self._precision = table[precision]
        if self._handle is not None and self._precision_symbol:
            _lib.check(getattr(_lib.lib(), self._precision_symbol)(self._handle, self._precision),
                       f"{type(self).__name__}.set_precision")

    def set_train_precision(self, precision) -> None:
        """Training-time arithmetic of the native forward_train / backward: "fp32" (default, the parity path) or "bf16" -- convolution
        operands rounded to bf16, fp32 accumulation, fp32 parameters / activations / gradients / optimiser (what Lightning's
        `precision: bf16-mixed` does to the reference's convolutions; include/dmel_hip.h: dmel_*_set_train_precision)."""
        table = {"fp32": 0, "float32": 0, torch.float32: 0, 0: 0, "32": 0, 32: 0, "bf16": 1, "bfloat16": 1, "bf16-mixed": 1, torch.bfloat16: 1, 1: 1}
        if precision not in table:
            raise ValueError(f"train precision must be 'fp32' or 'bf16', got {precision!r}")
        if table[precision] and not self._train_precision_symbol:
            raise NotImplementedError(f"{type(self).__name__} has no bf16 training mode")
        self._train_precision = table[precision]
        if self._handle is not None and self._train_precision_symbol:
            _lib.check(getattr(_lib.lib(), self._train_precision_symbol)(self._handle, self._train_precision),
                       f"{type(self).__name__}.set_train_precision")

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
