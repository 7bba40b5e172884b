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


def _grad_worker(rank: int, world: int, port: int, out):
    sys.path.insert(0, ROOT)
    from dmel_codec_amd.models.codec_lit_modules import VQGAN
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (7, 1000, 3, 50000)]       # several buckets at bucket_bytes = 4096
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        params.append(torch.nn.Parameter(torch.zeros(5)))                                # no gradient: skipped
        opt = torch.optim.SGD(params, lr=0.1)
        VQGAN.sync_gradients(opt, bucket_bytes=4096)
        out.put((rank, [float(p.grad[0]) for p in params[:4]], [float(p.grad.min()) == float(p.grad.max()) for p in params[:4]],
                 params[4].grad is None))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gradient_exchange_step_two_ranks():
    """VQGAN.sync_gradients (the one collective of the training path: bucketed all-reduce + average, RCCL on the GPUs) over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, firsts, uniform, untouched in res:
        assert firsts == [1.5 * (i + 1) for i in range(4)]        # mean of (1, 2) * (i + 1) on both ranks
        assert all(uniform) and untouched
