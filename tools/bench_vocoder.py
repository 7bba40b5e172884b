"""BigVGAN-only benchmark (BASELINE.json configs[3]: the 112-122 M parameter v2 vocoders, batch 16, 94 mel frames).

    python tools/bench_vocoder.py [--config v2_44k_128band_512x] [--batch 16] [--frames 94]
Prints one JSON line: audio-seconds per second, ms per forward, algorithmic conv TFLOP/s (serialised profiled pass)."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib
from dmel_codec_amd.configs import BIGVGAN, bigvgan_h
from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="v2_44k_128band_512x", choices=sorted(BIGVGAN))
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--frames", type=int, default=94)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
h = bigvgan_h(args.config)
torch.manual_seed(0)
m = BigVGAN(h)
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for name, p in m.named_parameters():
        if name.endswith("weight_v"):
            p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
        elif name.endswith("weight_g"):
            p.fill_(1.0)
m = m.to(dev)
mel = torch.randn(args.batch, h.num_mels, args.frames, device=dev)
for _ in range(2):
    y = m(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    y = m(mel)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
m.set_streams(1)
m(mel); torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(args.steps):
    m(mel)
torch.cuda.synchronize()
_lib.prof_enable(False)
conv, snake = _lib.prof_read("conv_igemm"), _lib.prof_read("aa_snake")
sr = {"base_24k_100band": 24000, "v2_24k_100band_256x": 24000, "v2_44k_128band_512x": 44100}[args.config]
print(json.dumps({"config": args.config, "params_M": round(sum(p.numel() for p in m.parameters()) / 1e6, 2),
                  "batch": args.batch, "frames": args.frames, "samples_out": y.shape[-1], "ms_per_forward": round(el * 1e3, 2),
                  "audio_sec_per_sec": round(args.batch * y.shape[-1] / sr / el, 1),
                  "conv_TFLOPs": round(conv["flops"] / conv["ms"] / 1e9, 1), "conv_ms": round(conv["ms"] / args.steps, 2),
                  "conv_gflop": round(conv["flops"] / args.steps / 1e9, 1),
                  "snake_ms": round(snake["ms"] / args.steps, 2), "snake_GBs": round(snake["bytes"] / snake["ms"] / 1e6, 1)}))
