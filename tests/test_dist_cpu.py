"""N>1 control path of bench.py on CPU: 2 ranks over gloo.  The data path has no collective (utterances shard across
ranks); what the ranks exchange is the barrier and the MAX of the elapsed time, and every rank draws its own clips."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank: int, world: int, port: int, out):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        audio = bench.synth_audio(4, 2400, 1234 + rank)          # each rank its own utterances
        dist.barrier()
        elapsed = bench.max_over_ranks(dist, 0.5 + rank, torch.device("cpu"))   # rank 1 is the slow one
        gathered = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(gathered, audio.abs().sum().view(1))
        out.put((rank, elapsed, [float(g) for g in gathered], bench.job_rate(world, 32, 1.0, 10, elapsed)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, sums0, rate0), (r1, e1, sums1, rate1) = res
    assert (r0, r1) == (0, 1)
    assert e0 == e1 == 1.5                       # max over ranks, seen by every rank
    assert sums0 == sums1 and sums0[0] != sums0[1]   # ranks hold different shards
    assert rate0 == rate1 == 2 * 32 * 1.0 * 10 / 1.5


def test_single_process_helpers():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.max_over_ranks(None, 0.25, torch.device("cpu")) == 0.25
    a = bench.synth_audio(2, 1000, 7)
    assert a.shape == (2, 1, 1000) and abs(float(a.abs().max()) - 0.95) < 1e-6
    assert torch.equal(a, bench.synth_audio(2, 1000, 7))


def _reducer_worker(rank: int, world: int, port: int, out):
    """A stand-in for a native module on the CPU: same bookkeeping (NativeModule._begin_train_call / _can_stream_grads /
    _deliver_grads), a backward that fills ONE flat gradient buffer block by block in reverse order and hands each block to the
    reducer as soon as it is complete -- exactly what dmel_wavenet_backward_hooked's on_ready callback does on the GPU."""
    sys.path.insert(0, ROOT)
    from dmel_codec_amd.ddp import GradReducer
    from dmel_codec_amd.models.modules._native import NativeModule
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L, n = 4, 1000
        events = []

        class Blocks(NativeModule):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(n)) for _ in range(L)])
                self.dead = torch.nn.Parameter(torch.zeros(3))      # like diffusion_projection: never receives a gradient

            def trained(self):
                return list(self.w)

            def forward(self, x):
                return _Fn.apply(self, x, *self.trained())

        class _Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, module, x, *params):
                ctx.module = module
                module._begin_train_call(ctx)
                ctx.save_for_backward(x)
                return x.sum() * 0 + sum((p * x[: p.numel()]).sum() for p in params)

            @staticmethod
            def backward(ctx, dy):
                module = ctx.module
                (x,) = ctx.saved_tensors
                module._check_train_call(ctx)
                params = module.trained()
                needs = ctx.needs_input_grad[2:]
                stream_out = module._can_stream_grads(params, needs)
                flat = torch.empty(L * n)
                for k in reversed(range(L)):                        # reverse layer order
                    flat[k * n:(k + 1) * n] = x[:n] * dy * (k + 1)
                    events.append(("computed", k))
                    if stream_out:
                        module._grad_sink.submit(flat[k * n:(k + 1) * n])
                grads = module._deliver_grads(flat, [(p, k * n, n) for k, p in enumerate(params)], needs, streamed=stream_out)
                return (None, None, *grads)

        m = Blocks()
        object.__setattr__(m, "_handle", 1)                          # "a native handle exists"
        extra = torch.nn.Linear(1, 2)                                # torch-native parameter (like quality_projection)
        unused = torch.nn.Parameter(torch.zeros(4))                  # no gradient on any rank: must stay None
        opt = torch.optim.SGD(list(m.parameters()) + list(extra.parameters()) + [unused], lr=0.1)
        red = GradReducer()
        red.record_events = True
        red.events = events
        x = torch.full((n,), float(rank + 1))
        # ---- pass 1: one backward through the module, per-block submission overlapped with the rest of backward
        red.arm([m])
        (m(x) + extra(torch.ones(1)).sum() * (rank + 1)).backward()
        red.finish(opt)
        order1 = list(events)
        g1 = [float(p.grad[0]) for p in m.w]
        uniform1 = all(float(p.grad.min()) == float(p.grad.max()) for p in m.w)
        extra_g = [float(extra.weight.grad[0, 0]), float(extra.bias.grad[1])]
        # ---- pass 2: gradients already present + two uses of the module in one graph (the discriminator on real and fake mels)
        events.clear()
        red.arm([m])
        (m(x) + 2 * m(x)).backward()
        red.finish(opt)
        order2 = list(events)
        g2 = [float(p.grad[0]) for p in m.w]
        # ---- unarmed: plain autograd accumulation (single-rank behaviour)
        opt.zero_grad()
        m(x).backward()
        g3 = [float(p.grad[0]) for p in m.w]
        # ---- robustness of the exchange
        from dmel_codec_amd.ddp import broadcast_parameters, check_parameters_in_sync
        robust = {}
        # (a) a grad-enabled forward whose graph is dropped without backward (an encode() in a callback) must not block later exchanges
        opt.zero_grad()
        dropped = m(x)
        del dropped
        events.clear()
        with red.exchange([m], opt):
            m(x).backward()
        robust["dropped_graph_issues"] = [e for e in events if e[0] == "issue"]
        robust["dropped_graph_grad"] = float(m.w[0].grad[0])
        # (b) a LIVE forward without backward: the module's gradients cannot leave -> finish() raises instead of letting the ranks diverge
        opt.zero_grad()
        alive = m(x)
        red.arm([m])
        m(x).backward()
        try:
            red.finish(opt)
            robust["stuck_raises"] = False
        except RuntimeError as e:
            robust["stuck_raises"] = "not exchanged" in str(e)
        robust["disarmed_after_stuck"] = m._grad_sink is None
        del alive
        # (c) an exception inside the backward of an exchange() block disarms the modules
        opt.zero_grad()
        try:
            with red.exchange([m], opt):
                raise ValueError("backward failed")
        except ValueError:
            pass
        robust["disarmed_after_error"] = m._grad_sink is None and not red._armed and not red._pending
        # (d) the RCCL branch (ReduceOp.AVG inside the collective, no scale pass afterwards), with a stub: gloo has no AVG, so the stub
        #     checks that AVG was asked for and performs SUM / world in its place
        import dmel_codec_amd.ddp as ddp_mod
        real_all_reduce, real_backend = dist.all_reduce, dist.get_backend
        asked = []

        class _DoneWork:
            def wait(self):
                return True

        def fake_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
            asked.append(op)
            if op == dist.ReduceOp.AVG:
                real_all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                t.div_(world)
                return _DoneWork() if async_op else None
            return real_all_reduce(t, op=op, group=group, async_op=async_op)

        ddp_mod.dist.all_reduce, ddp_mod.dist.get_backend = fake_all_reduce, (lambda group=None: "nccl")
        try:
            opt.zero_grad()
            with red.exchange([m], opt):
                m(x).backward()
        finally:
            ddp_mod.dist.all_reduce, ddp_mod.dist.get_backend = real_all_reduce, real_backend
        robust["avg_ops"] = sum(1 for o in asked if o == dist.ReduceOp.AVG)
        robust["avg_grad"] = [float(p.grad[0]) for p in m.w]
        # (e) rank 0's parameters become everyone's; a checksum mismatch is reported
        with torch.no_grad():
            for p in m.w:
                p.fill_(float(rank) + 0.25)
        try:
            check_parameters_in_sync(m)
            robust["mismatch_detected"] = False
        except RuntimeError:
            robust["mismatch_detected"] = True
        robust["n_broadcast"] = broadcast_parameters(m)
        check_parameters_in_sync(m)
        robust["after_broadcast"] = [float(p[0]) for p in m.w]
        out.put((rank, order1, g1, uniform1, extra_g, unused.grad is None, m.dead.grad is None, order2, g2, g3, m._pending_train, robust))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_overlaps_backward_two_ranks():
    """dmel_codec_amd.ddp.GradReducer over gloo: per-block in-place all-reduce issued from inside backward in reverse layer order (a
    block's collective is issued BEFORE the remaining blocks are differentiated), averaged values, static collectives (a parameter
    without gradient on every rank keeps None), accumulation into existing gradients, modules used twice in one graph."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    L, n = 4, 1000
    for rank, order1, g1, uniform1, extra_g, unused_none, dead_none, order2, g2, g3, pending, robust in res:
        assert robust["dropped_graph_issues"] == [("issue", n)] * L and robust["dropped_graph_grad"] == 1.5
        assert robust["stuck_raises"] and robust["disarmed_after_stuck"] and robust["disarmed_after_error"]
        assert robust["avg_ops"] == L and robust["avg_grad"] == [1.5 * (k + 1) for k in range(L)]
        assert robust["mismatch_detected"] and robust["n_broadcast"] == L + 1 and robust["after_broadcast"] == [0.25] * L
        # pass 1: computed(3), issue, computed(2), issue, ... : block k is on the wire before block k-1 is differentiated
        assert order1[0] == ("arm", 1)
        body = order1[1:1 + 2 * L]
        assert body == [e for k in reversed(range(L)) for e in (("computed", k), ("issue", n))], body
        assert order1[-2][0] == "issue_rest" and order1[-1] == ("finish", L)
        assert g1 == [1.5 * (k + 1) for k in range(L)] and uniform1          # mean over ranks of (rank + 1) * (k + 1)
        assert extra_g == [1.5, 1.5] and unused_none and dead_none
        # pass 2: nothing leaves until the module's last outstanding backward; then ONE message holding old + both new gradients
        issues = [e for e in order2 if e[0] == "issue"]
        assert issues == [("issue", L * n)]
        assert order2.index(("issue", L * n)) > max(i for i, e in enumerate(order2) if e[0] == "computed")
        assert g2 == [1.5 * (k + 1) * 4 for k in range(L)]                   # avg(old) + avg(1 + 2 uses)
        assert g3 == [float(rank + 1) * (k + 1) for k in range(L)] and pending == 0
