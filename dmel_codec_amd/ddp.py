"""Data-parallel gradient exchange for `train_codec.py`: one process per GPU, RCCL all-reduce over xGMI, overlapped with backward.

The reference gets this from Lightning's DDP wrapper (`strategy: ddp_find_unused_parameters_true`, config/codec/dMel_example.yaml:14;
two reductions per step, after `manual_backward(loss_d)` and `manual_backward(loss)`, codec_lit_modules.py:236,315).  Here the native
backward of every module writes ONE flat fp32 gradient buffer (include/dmel_hip.h: dmel_*_grad_floats / _grad_slot); while a
GradReducer is armed on a module its parameters' `.grad` are views of that buffer and the buffer is all-reduced IN PLACE -- no
flatten / copy-back passes.  The decoder WaveNet hands its buffer over block by block from inside the native backward
(dmel_wavenet_backward_hooked), so block k's all-reduce is in flight while blocks k-1 ... 0 are still being differentiated
(one bucket per WaveNet block, reverse layer order).

Robustness: outstanding training forwards are tracked by weak references to their autograd contexts, so a graph dropped
without backward cannot block later exchanges; `finish()` raises if a module produced gradients that never left; `exchange()` is the
try/finally form of arm / backward / finish; `broadcast_parameters()` makes rank 0's weights everyone's before the first step and
`check_parameters_in_sync()` compares a checksum across ranks.  The RCCL transport (backend "nccl", ReduceOp.AVG) has run with ONE rank only
(tests/test_gpu_train.py::test_training_step_over_rccl_single_rank: all code on this side of the wire, no peer); every multi-rank test
uses gloo.

Every rank issues the same collectives in the same order by construction: the order is the order in which the (static) module graph is
walked backwards, never which `.grad` happens to be None on a rank.  torch.distributed's NCCL (= RCCL) process group runs a collective
on its own stream after the work already enqueued on the current stream, and `Work.wait()` makes the current stream wait for it: that
is the side stream + event pair of the design, provided by the backend.  Under gloo (CPU tests) the same calls run on host threads."""
from __future__ import annotations

import contextlib
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _exchanging(group=None) -> bool:
    """Is there anybody to exchange with?  A one-rank group normally skips every collective; DMEL_DDP_EXERCISE_SINGLE_RANK=1 runs them
    anyway (tools/ddp_rehearsal.py on a one-GPU box: RCCL wants one device per rank, so the only way to put the RCCL code path -- AVG,
    communicator stream, Work handles -- under test there is a group of one)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DMEL_DDP_EXERCISE_SINGLE_RANK") == "1"


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
        return _exchanging(self.group)

    def _avg_op(self):
        # RCCL averages inside the collective; gloo has no AVG: sum, then one scale pass after the wait
        return dist.get_backend(self.group) == "nccl"

    def arm(self, modules: Iterable[torch.nn.Module]) -> None:
        """Route the native gradient buffers of `modules` (and their NativeModule children) through this reducer for the next backward
        pass.  No-op for a single rank: the modules keep handing their gradients to autograd."""
        from .models.modules._native import NativeModule
        if self._armed or self._pending:
            # a previous pass was abandoned between arm() and finish() (an exception in backward): disarm its modules and join its
            # collectives before anything new is issued, so that the ranks' collective sequences cannot interleave
            self.abort()
        if not self.active:
            return
        for m in modules:
            if m is None:
                continue
            for sub in m.modules():
                if isinstance(sub, NativeModule):
                    sub._grad_sink = self
                    sub._grads_delivered = sub._grads_submitted = False
                    self._armed.append(sub)
        if self.record_events:
            self.events.append(("arm", len(self._armed)))

    def abort(self) -> None:
        """Disarm every module and wait for whatever was already issued, without touching gradients any further.  For error paths:
        the collectives a rank has issued must still complete on every rank."""
        for sub in self._armed:
            sub._grad_sink = None
        self._armed = []
        pending, self._pending = self._pending, []
        for work, _, _ in pending:
            try:
                work.wait()
            except Exception:       # the process group may already be broken: nothing more to do for this message
                pass

    @contextlib.contextmanager
    def exchange(self, modules: Iterable[torch.nn.Module], optimizer: Optional[torch.optim.Optimizer] = None):
        """`with reducer.exchange([modules], optimizer): loss.backward()` = arm, backward, finish -- and, if backward raises, disarm and
        join what was issued before the exception propagates (an armed module left behind would send the next step's gradients into
        a reducer whose collective order no longer matches the other ranks')."""
        self.arm(modules)
        try:
            yield self
        except BaseException:
            self.abort()
            raise
        self.finish(optimizer)

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
        stuck = []
        for sub in self._armed:
            sub._grad_sink = None
            for p in sub.parameters(recurse=True):
                native_owned.add(id(p))
            # gradients were pointed at the module's flat buffer but the buffer never left: some OTHER training forward of this module is
            # still waiting for a backward that did not run in this pass.  Silently skipping the exchange would let the ranks diverge.
            if sub._grads_delivered and not sub._grads_submitted:
                stuck.append(f"{type(sub).__name__} ({sub._pending_train} training forward(s) without backward)")
        self._armed = []
        if stuck:
            self.abort()
            raise RuntimeError("GradReducer.finish: gradients of " + ", ".join(stuck) + " were produced but not exchanged: a grad-enabled "
                               "forward of the module is still alive without its backward (run such calls under torch.no_grad(), or "
                               "drop their outputs before backward)")
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


def _flat_state(module: torch.nn.Module):
    return [t for t in list(module.parameters()) + list(module.buffers()) if t is not None and t.numel() > 0]


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> int:
    """Rank `src`'s parameters and buffers become every rank's (what Lightning's DDP wrapper does when it is constructed;
    train_codec.py:49-55 relies on it): identical seeds are then a convenience, not the thing consistency rests on.  Floating tensors
    travel in chunks of one flat message per dtype; returns the number of tensors sent."""
    if not _exchanging(group):
        return 0
    tensors = _flat_state(module)
    by_kind: dict = {}
    for t in tensors:
        by_kind.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), ts in by_kind.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view(t.shape))        # in place: versions move, native handles re-pack on next use
            off += n
    return len(tensors)


@torch.no_grad()
def check_parameters_in_sync(module: torch.nn.Module, group=None, what: str = "parameters") -> None:
    """Raise if the ranks hold different parameters: every rank contributes (sum, sum of squares, count) of its floating tensors in
    float64 and the MIN and MAX over ranks must agree exactly (bit-identical weights give bit-identical sums)."""
    if not _exchanging(group):
        return
    tensors = [t for t in _flat_state(module) if t.is_floating_point()]
    dev = tensors[0].device if tensors else torch.device("cpu")
    s = torch.zeros(3, dtype=torch.float64, device=dev)
    for t in tensors:
        d = t.detach().double()
        s[0] += d.sum()
        s[1] += (d * d).sum()
        s[2] += t.numel()
    lo, hi = s.clone(), s.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"data-parallel ranks hold different {what}: checksum min {lo.tolist()} != max {hi.tolist()} "
                           "(gradient exchange skipped somewhere, or the ranks started from different weights)")
