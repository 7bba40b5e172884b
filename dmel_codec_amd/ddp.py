"""Data-parallel gradient exchange for `train_codec.py`: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference gets this from Lightning's DDP wrapper (`strategy: ddp_find_unused_parameters_true`, config/codec/dMel_example.yaml:14;
two reductions per step, after `manual_backward(loss_d)` and `manual_backward(loss)`, codec_lit_modules.py:236,315).  Here the native
backward of every module writes ONE flat fp32 gradient buffer (include/dmel_hip.h: dmel_*_grad_floats / _grad_slot); while a
GradReducer is armed on a module its parameters' `.grad` are views of that buffer and the buffer is all-reduced IN PLACE -- no
flatten / copy-back passes.  The decoder WaveNet hands its buffer over block by block from inside the native backward
(dmel_wavenet_backward_hooked), so block k's all-reduce is in flight while blocks k-1 ... 0 are still being differentiated
(one bucket per WaveNet block, reverse layer order).

Every rank issues the same collectives in the same order by construction: the order is the order in which the (static) module graph is
walked backwards, never which `.grad` happens to be None on a rank.  torch.distributed's NCCL (= RCCL) process group runs a collective
on its own stream after the work already enqueued on the current stream, and `Work.wait()` makes the current stream wait for it: that
is the side stream + event pair of the design, provided by the backend.  Under gloo (CPU tests) the same calls run on host threads."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, max_message_bytes: int = 256 << 20):
        self.group = group
        self.max_message_bytes = int(max_message_bytes)
        self._pending: List[tuple] = []       # (work, tensor, divide_after)
        self._armed: List = []
        self.events: List[tuple] = []         # ("arm", n) / ("issue", numel) / ("finish", n_collectives): order of issue, for tests
        self.record_events = False

    # ---------------------------------------------------------------------------------------------------------
    @property
    def active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _avg_op(self):
        # RCCL averages inside the collective; gloo has no AVG: sum, then one scale pass after the wait
        return dist.get_backend(self.group) == "nccl"

    def arm(self, modules: Iterable[torch.nn.Module]) -> None:
        """Route the native gradient buffers of `modules` (and their NativeModule children) through this reducer for the next backward
        pass.  No-op for a single rank: the modules keep handing their gradients to autograd."""
        from .models.modules._native import NativeModule
        self._armed = []
        if not self.active:
            return
        for m in modules:
            if m is None:
                continue
            for sub in m.modules():
                if isinstance(sub, NativeModule):
                    sub._grad_sink = self
                    self._armed.append(sub)
        if self.record_events:
            self.events.append(("arm", len(self._armed)))

    def submit(self, flat: torch.Tensor) -> None:
        """All-reduce (average) a contiguous 1-D fp32 gradient region in place, asynchronously.  Called in backward order."""
        assert flat.is_contiguous() and flat.ndim == 1
        step = max(1, self.max_message_bytes // flat.element_size())
        for lo in range(0, flat.numel(), step):
            piece = flat[lo:lo + step]
            if self._avg_op():
                work = dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                self._pending.append((work, piece, False))
            else:
                work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append((work, piece, True))
            if self.record_events:
                self.events.append(("issue", piece.numel()))

    def finish(self, optimizer: Optional[torch.optim.Optimizer] = None) -> None:
        """Disarm; exchange the gradients of `optimizer`'s remaining (non-native, e.g. quality_projection) parameters in one small
        message; wait for every collective of this pass.  After this the current stream sees averaged gradients everywhere."""
        from .models.modules._native import NativeModule
        native_owned = set()
        for sub in self._armed:
            sub._grad_sink = None
            for p in sub.parameters(recurse=True):
                native_owned.add(id(p))
        self._armed = []
        if not self.active:
            return
        world = dist.get_world_size(self.group)
        rest = []
        if optimizer is not None:
            rest = [p for grp in optimizer.param_groups for p in grp["params"] if p.requires_grad and id(p) not in native_owned]
        if rest:
            # static message: every parameter's gradient (zeros where this rank has none) + one "has a gradient" flag per parameter,
            # so that a parameter unused on EVERY rank keeps grad None (what find_unused_parameters gives the reference) without the
            # ranks having to agree on the message layout beforehand
            dev = rest[0].device
            parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in rest]
            flags = torch.tensor([0.0 if p.grad is None else 1.0 for p in rest], device=dev)
            msg = torch.cat(parts + [flags])
            dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=self.group)
            used = msg[-len(rest):].tolist()
            off = 0
            for p, u in zip(rest, used):
                n = p.numel()
                if u > 0:
                    g = (msg[off:off + n] / world).view(p.shape).to(p.dtype)
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.copy_(g)
                off += n
            if self.record_events:
                self.events.append(("issue_rest", int(msg.numel())))
        n = len(self._pending)
        for work, piece, divide in self._pending:
            work.wait()
            if divide:
                piece.div_(world)
        self._pending = []
        if self.record_events:
            self.events.append(("finish", n))
        del NativeModule
