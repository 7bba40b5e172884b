"""Whole-WaveNet forward micro-benchmark (inference path): decoder / encoder of the bench configuration, hipEvent-timed.
    python tools/bench_wavenet.py [--which dec|enc] [--batch 32] [--frames 92] [--reps 30]
A/B switches are environment variables read by the library: DMEL_WAVENET_FOLD=0|1, DMEL_CONV_TILE_BF16=0..5."""
import argparse
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib
from dmel_codec_amd.models.modules.wavenet import WaveNet

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="dec")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=92)
ap.add_argument("--groups", type=int, default=8)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--precision", default=None, help="fp32 (six-product split) | fp32_f16x2 (what VQGAN asks of its decoder); default: f16x2 for dec, fp32 for enc")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
torch.set_grad_enabled(False)        # the inference path (with gradients enabled the mirrors run their training forward)
C = 70 * args.groups
if args.which == "dec":
    m = WaveNet(input_channels=C, output_channels=10 * args.groups, residual_channels=C, residual_layers=20, dilation_cycle=4,
                condition_channels=C).to(dev)
    N = args.batch
    x = torch.randn(N, C, args.frames, device=dev)
    c = torch.randn(N, C, args.frames, device=dev)
    flops = 2.0 * N * args.frames * (20 * (2 * C * (3 * C + C) + 2 * C * C) + C * C + C * 10 * args.groups)
else:
    m = WaveNet(input_channels=10, residual_channels=70, residual_layers=20, dilation_cycle=4).to(dev)
    N = args.batch * args.groups
    x = torch.randn(N, 10, args.frames, device=dev)
    c = None
    flops = 2.0 * N * args.frames * (10 * 70 + 20 * (140 * 210 + 140 * 70) + 70 * 70)
m.set_precision(args.precision or ("fp32_f16x2" if args.which == "dec" else "fp32"))
for _ in range(3):
    y = m(x, condition=c)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(args.reps):
    y = m(x, condition=c)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / args.reps
_lib.prof_reset(); _lib.prof_enable(True)
y = m(x, condition=c)
torch.cuda.synchronize()
_lib.prof_enable(False)
conv = _lib.prof_read("conv_igemm")
print(f"{args.which} N={N} T={args.frames} fold={os.environ.get('DMEL_WAVENET_FOLD', 'auto')} tile={os.environ.get('DMEL_CONV_TILE_BF16', 'auto')}: "
      f"{ms:.3f} ms/forward = {flops / ms / 1e9:.1f} TF/s; conv launches {conv['launches']} avg {1e3 * conv['ms'] / max(1, conv['launches']):.1f} us; "
      f"finite={bool(torch.isfinite(y).all())}")
