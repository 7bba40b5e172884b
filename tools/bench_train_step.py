"""Full VQGAN.training_step (codec_lit_modules.py:159-327: discriminator + generator steps, AdamW, schedulers, device-side weight
re-pack) on the native training paths, BASELINE cfg-2/3 shapes.  GPU only.  Under `python -m torch.distributed.run --nproc-per-node N
tools/bench_train_step.py` every rank trains on its own clips and the gradients are exchanged with RCCL (dmel_codec_amd.ddp.GradReducer,
overlapped with backward).  --precision bf16 selects the bf16 training mode.

    python tools/bench_train_step.py [--batch 32] [--seconds 1.0] [--steps 5]
"""
import argparse, json, os, sys, time
from functools import partial
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmel_codec_amd import _lib
from dmel_codec_amd.configs import build_codec
from dmel_codec_amd.utils.schedule import get_cosine_schedule_with_warmup_lr_lambda

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"))
args = ap.parse_args()
world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(114514)
opt = partial(torch.optim.AdamW, lr=1e-4, betas=(0.8, 0.99), eps=1e-5)                  # dMel_example.yaml optimizer / scheduler blocks
sched = partial(torch.optim.lr_scheduler.LambdaLR, lr_lambda=partial(get_cosine_schedule_with_warmup_lr_lambda, num_warmup_steps=100,
                                                                      num_training_steps=1000000, final_lr_ratio=0.0))
codec = build_codec(n_mels=80, dmel_groups=8, vocoder=None, discriminator=True, optimizer=opt, lr_scheduler=sched).to(dev)
codec.set_train_precision(args.precision)
L = int(24000 * args.seconds)
audio = bench.synth_audio(args.batch, L, 1234 + rank).to(dev)
lens = torch.full((args.batch,), L, device=dev, dtype=torch.int64)
batch = {"audios": audio, "audio_lengths": lens}
for i in range(2):
    logs = codec.training_step(batch, i)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for i in range(args.steps):
    logs = codec.training_step(batch, 2 + i)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
_lib.prof_reset(); _lib.prof_enable(True)
codec.training_step(batch, 99); torch.cuda.synchronize()
_lib.prof_enable(False)
fam = {k: round(_lib.prof_read(k)["ms"], 2) for k in ("conv_igemm", "conv_wgrad", "train_elementwise", "small")}
if rank == 0:
    n_g = sum(p.numel() for k, p in codec.named_parameters() if not k.startswith("discriminator.") and "diffusion_projection" not in k)
    n_d = sum(p.numel() for p in codec.discriminator.parameters())
    print(json.dumps({"workload": f"VQGAN.training_step, {world} rank(s) x {args.batch} x {args.seconds:g} s @24 kHz, 80 mel / 8 groups, WaveNet 20+20 "
                                  f"({n_g / 1e6:.1f} M) + discriminator ({n_d / 1e6:.1f} M)",
                      "precision": args.precision, "ms_per_step": round(el * 1e3, 2), "audio_sec_per_sec": round(world * args.batch * args.seconds / el, 1),
                      "kernel_ms_one_step": fam, "losses": {k: round(v, 4) for k, v in logs.items()}}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
