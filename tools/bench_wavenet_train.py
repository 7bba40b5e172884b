"""Training-path benchmark of the decoder WaveNet (BASELINE cfg-3's hot loop: 20 conditioned layers, 560 channels): native
forward_train + backward per step, with the per-family kernel times from the library's hipEvent hooks (GPU only).

    python tools/bench_wavenet_train.py [--batch 32] [--frames 92] [--layers 20] [--channels 560] [--steps 5]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib
from dmel_codec_amd.models.modules.wavenet import WaveNet

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--frames", type=int, default=92)
ap.add_argument("--layers", type=int, default=20)
ap.add_argument("--channels", type=int, default=560)
ap.add_argument("--out", type=int, default=80)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
C = args.channels
m = WaveNet(input_channels=C, output_channels=args.out, residual_channels=C, residual_layers=args.layers, dilation_cycle=4,
            condition_channels=C).to(dev)
x = torch.randn(args.batch, C, args.frames, device=dev)
cond = torch.randn(args.batch, C, args.frames, device=dev, requires_grad=True)
gy = torch.randn(args.batch, args.out, args.frames, device=dev)


def step():
    m.zero_grad(set_to_none=True)
    y = m(x, condition=cond)
    (y * gy).sum().backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.steps
_lib.prof_reset(); _lib.prof_enable(True)
step(); torch.cuda.synchronize()
_lib.prof_enable(False)
fam = {k: _lib.prof_read(k) for k in ("conv_igemm", "conv_wgrad", "train_elementwise")}
_lib.prof_reset()
with torch.no_grad():
    for p_ in m.parameters():
        p_.add_(0.0)                      # bump the versions like an optimiser step would
torch.cuda.synchronize()
t1 = time.perf_counter()
m.native()
torch.cuda.synchronize()
repack_s = time.perf_counter() - t1
params = sum(p.numel() for k, p in m.named_parameters() if "diffusion_projection" not in k)
# forward 2*MACs; backward-data the same again (minus the first layer's input), backward-weight the same again
macs = args.batch * args.frames * (args.layers * (2 * C * C * 3 + 2 * C * C + 2 * C * C) + C * C + C * args.out)
print(json.dumps({
    "workload": f"decoder WaveNet training step (forward_train + backward), batch {args.batch} x {args.frames} frames, "
                f"{args.layers} layers x {C} channels, {params / 1e6:.1f} M trained parameters",
    "ms_per_step": round(el * 1e3, 3), "frames_per_sec": round(args.batch * args.frames / el, 1),
    "algorithmic_tflops_fwd_plus_bwd": round(6.0 * macs / el / 1e12, 1),
    "kernel_ms": {k: round(v["ms"], 3) for k, v in fam.items()},
    "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 and v["flops"] > 0 else None for k, v in fam.items()},
    "device_repack_after_optimizer_step_ms": round(repack_s * 1e3, 2),
    "note": "parameter versions do not change between the timed steps; after an optimiser step every weight image is re-packed "
            "on the device from the live parameters (dmel_wavenet_refresh: device_repack_after_optimizer_step_ms, not in ms_per_step)"}))
