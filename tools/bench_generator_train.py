"""Generator half of training_step (encoder WaveNet -> straight-through quantiser -> conditioned decoder WaveNet -> band-weighted mel
L1, codec_lit_modules.py:164-211, 246-263) forward + backward on the native training paths, BASELINE cfg-2 shapes (GPU only).

    python tools/bench_generator_train.py [--batch 32] [--seconds 1.0] [--steps 5]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmel_codec_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
codec = bench.build("cfg2").to(dev)
codec.vocoder = None
L = int(24000 * args.seconds)
audio = bench.synth_audio(args.batch, L, 1234).to(dev)
lens = torch.full((args.batch,), L, device=dev, dtype=torch.int64)
params = [p for k, p in codec.named_parameters() if "diffusion_projection" not in k]
opt = torch.optim.AdamW(params, lr=1e-4, betas=(0.8, 0.99), eps=1e-5)


def step(do_opt):
    opt.zero_grad(set_to_none=True)
    gen_mel, gt, m = codec.generator_forward(audio, lens)
    loss = codec.mel_loss(gen_mel, gt, m)
    loss.backward()
    if do_opt:
        opt.step()
    return loss


for _ in range(2):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step(False)
torch.cuda.synchronize()
fb = (time.perf_counter() - t0) / args.steps
for _ in range(2):
    step(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss = step(True)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / args.steps
_lib.prof_reset(); _lib.prof_enable(True)
step(False); torch.cuda.synchronize()
_lib.prof_enable(False)
fam = {k: round(_lib.prof_read(k)["ms"], 3) for k in ("conv_igemm", "conv_wgrad", "train_elementwise", "small", "stft_logmel")}
print(json.dumps({
    "workload": f"generator half of training_step: {args.batch} x {args.seconds:g} s @24 kHz, 80 mel / 8 groups, WaveNet 20+20, "
                f"{sum(p.numel() for p in params) / 1e6:.1f} M trained parameters",
    "forward_backward_ms": round(fb * 1e3, 2), "audio_sec_per_sec_forward_backward": round(args.batch * args.seconds / fb, 1),
    "with_adamw_and_device_repack_ms": round(full * 1e3, 2), "audio_sec_per_sec_full_step": round(args.batch * args.seconds / full, 1),
    "kernel_ms_forward_backward": fam, "loss": round(float(loss), 4)}))
