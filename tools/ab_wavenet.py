"""Interleaved A/B of decoder-WaveNet launch configurations inside ONE process on ONE device (boxes differ by 10-15 %, and so do
consecutive runs on one box: only interleaved rounds compare).  Prints the median of `rounds` timings per configuration."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd.models.modules.wavenet import WaveNet

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
C = 70 * groups
m = WaveNet(input_channels=C, output_channels=10 * groups, residual_channels=C, residual_layers=20, dilation_cycle=4, condition_channels=C).to(dev)
x = torch.randn(batch, C, 92, device=dev)
c = torch.randn(batch, C, 92, device=dev)
configs = [("per-item, auto tile", {}), ("folded, 64x128", {"DMEL_WAVENET_FOLD": "1", "DMEL_CONV_TILE_BF16": "2"}),
           ("per-item, 64x128", {"DMEL_CONV_TILE_BF16": "2"}), ("folded, 128x96", {"DMEL_WAVENET_FOLD": "1", "DMEL_CONV_TILE_BF16": "1"}),
           ("folded, 128x128 (2x2 waves)", {"DMEL_WAVENET_FOLD": "1", "DMEL_CONV_TILE_BF16": "0"})]
times = {name: [] for name, _ in configs}
for rnd in range(7):
    for name, env in configs:
        for k in ("DMEL_WAVENET_FOLD", "DMEL_CONV_TILE_BF16"):
            os.environ.pop(k, None)
        os.environ.update(env)
        m(x, condition=c)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            y = m(x, condition=c)
        b.record()
        torch.cuda.synchronize()
        times[name].append(a.elapsed_time(b) / 10)
for name, _ in configs:
    t = times[name]
    print(f"{name:32s} median {statistics.median(t):.3f} ms  (min {min(t):.3f}, max {max(t):.3f})")
