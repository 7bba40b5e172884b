"""Two (or more) data-parallel ranks of the REAL training step on ONE GPU: a rehearsal of train_codec.py's exchange step for boxes with
a single device.  RCCL refuses two ranks on one device, so the process group is gloo (it moves CUDA tensors through the host); what is
exercised is everything else: the native backward handing its flat gradient buffer over block by block (dmel_wavenet_backward_hooked),
parameter gradients that are views of that buffer, asynchronous in-place all-reduces issued from inside backward, the wait before the
clip, static collectives.  Every rank trains on its own clips; afterwards all ranks must hold IDENTICAL parameters, and those must equal
what a single process computes from the ranks' gradients averaged by hand.

    python tools/ddp_rehearsal.py            # parent: starts WORLD ranks of itself, prints PASS / FAIL
"""
import os, subprocess, sys, json
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WORLD = int(os.environ.get("DMEL_REHEARSAL_WORLD", "2"))
# gloo (default: several ranks on one device) or nccl = RCCL (one rank per device: on a one-GPU box that is WORLD = 1 -- the transport the
# 8-GPU run uses, with its AVG reduction, communicator stream and Work handles, just without a peer)
BACKEND = os.environ.get("DMEL_REHEARSAL_BACKEND", "gloo")


def build():
    from functools import partial
    from dmel_codec_amd.configs import build_codec
    torch.manual_seed(7)
    opt = partial(torch.optim.AdamW, lr=1e-3, betas=(0.8, 0.99), eps=1e-5)
    sched = partial(torch.optim.lr_scheduler.LambdaLR, lr_lambda=lambda s: 1.0)
    codec = build_codec(n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3, vocoder=None, discriminator=True, optimizer=opt,
                        lr_scheduler=sched)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n, p in codec.named_parameters():
            if p.ndim >= 2 and "discriminator" not in n:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
    return codec


def batch_for(rank, step, dev):
    g = torch.Generator().manual_seed(1000 * rank + step)
    audio = (torch.randn(2, 1, 12000, generator=g) * 0.2).to(dev)
    lens = torch.tensor([12000, 9000 - 500 * rank], device=dev)
    noise = torch.randn(2, 560, 12000 // 256, generator=g).to(dev)
    return {"audios": audio, "audio_lengths": lens}, noise


def worker():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if BACKEND == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    codec = build().to(dev)
    codec.grad_reducer.record_events = True
    logs = []
    for step in range(2):
        b, noise = batch_for(rank, step, dev)
        logs.append(codec.training_step(b, step, noise=noise))
    torch.cuda.synchronize()
    ev = codec.grad_reducer.events
    issues = [e for e in ev if e[0] == "issue"]
    flat = torch.cat([p.detach().reshape(-1).double().cpu() for p in codec.parameters()])
    digest = [float(flat.sum()), float(flat.abs().sum()), float((flat * torch.arange(flat.numel(), dtype=torch.float64)).sum())]
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    same = all(g == gathered[0] for g in gathered)
    if rank == 0:
        torch.save({k: v.detach().cpu() for k, v in codec.state_dict().items()}, os.environ["DMEL_REHEARSAL_OUT"])
        print(json.dumps({"backend": dist.get_backend(), "world": world, "ranks_identical": same, "n_collectives": len(issues), "arms": sum(1 for e in ev if e[0] == "arm"),
                          "loss_mel_rank0": [l["train/generator/loss_mel"] for l in logs]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if same else 3)


def reference(out_path):
    """single process: every rank's gradients computed one after the other on copies of the same model, averaged by hand"""
    dev = torch.device("cuda:0")
    models = [build().to(dev) for _ in range(WORLD)]
    from dmel_codec_amd.utils.utils import avg_with_mask
    import torch.nn.functional as F
    for step in range(2):
        # the two halves of training_step (codec_lit_modules.py:213-244, 246-327) with the replicas in lock step
        outs = []
        for r, m in enumerate(models):
            b, noise = batch_for(r, step, dev)
            gen_mel, gt, mask = m.generator_forward(b["audios"], b["audio_lengths"], noise=noise)
            real, fake = m.discriminator(gt), m.discriminator(gen_mel.detach())
            dmask = F.interpolate(mask, size=(real.shape[2],), mode="nearest")
            (avg_with_mask((real - 1) ** 2, dmask) + avg_with_mask(fake ** 2, dmask)).backward()
            outs.append((gen_mel, gt, mask, dmask))
        _average([list(m.discriminator.parameters()) for m in models])
        for m in models:
            od = m.optimizers()[1]
            m.clip_gradients(od, 1000.0); od.step(); od.zero_grad(); m.lr_schedulers()[1].step()
        for m, (gen_mel, gt, mask, dmask) in zip(models, outs):
            loss = m.weight_mel * m.mel_loss(gen_mel, gt, mask) + m.weight_adv * avg_with_mask((m.discriminator(gen_mel) - 1) ** 2, dmask)
            loss.backward()
        gparams = lambda m: [p for grp in m.optimizers()[0].param_groups for p in grp["params"]]
        _average([gparams(m) for m in models])
        for m in models:
            og = m.optimizers()[0]
            m.clip_gradients(og, 1000.0); og.step(); og.zero_grad(); m.lr_schedulers()[0].step()
    got = torch.load(out_path)
    want = {k: v.detach().cpu() for k, v in models[0].state_dict().items()}
    worst = 0.0
    for k, v in want.items():
        if v.is_floating_point():
            worst = max(worst, float((got[k].double() - v.double()).abs().max() / v.double().abs().max().clamp(min=1e-12)))
    return worst


def _average(param_lists):
    for ps in zip(*param_lists):
        have = [p.grad for p in ps if p.grad is not None]
        if not have:
            continue
        mean = sum((p.grad if p.grad is not None else torch.zeros_like(p)) for p in ps) / len(ps)
        for p in ps:
            p.grad = mean.clone()


def main():
    if "RANK" in os.environ:
        return worker()
    import socket, tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "rank0.pt")
    procs = []
    for r in range(WORLD):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DMEL_REHEARSAL_OUT=out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    try:
        rcs = [p.wait(timeout=int(os.environ.get("DMEL_REHEARSAL_TIMEOUT", "600"))) for p in procs]
    except subprocess.TimeoutExpired:
        for p in procs:      # exactly the processes started above
            p.kill()
        print("FAIL: a rank did not finish in time")
        sys.exit(1)
    if any(rcs):
        print(f"FAIL: rank exit codes {rcs}")
        sys.exit(1)
    worst = reference(out)
    print(f"reference (gradients averaged by hand in one process): worst relative parameter difference {worst:.2e}")
    ok = worst < 5e-3
    print("PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
