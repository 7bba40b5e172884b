"""STFT->log-mel kernel at sizes where it leaves the launch-latency regime (GPU only): GB/s of algorithmic traffic.

    python tools/bench_stft.py [--one]      # --one: only 256 x 60 s, 3 launches (for rocprofv3 --pmc passes)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram

dev = torch.device("cuda:0")
m = LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=80)
ONE = "--one" in sys.argv
for B, secs in (((256, 60),) if ONE else ((32, 1), (32, 10), (32, 60), (256, 60))):
    x = torch.randn(B, 24000 * secs, device=dev) * 0.1
    for _ in range(1 if ONE else 3):
        y = m(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    NIT = 2 if ONE else 10
    for _ in range(NIT):
        y = m(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / NIT
    nbytes = 4.0 * x.numel() + 4.0 * y.numel()
    frames = y.shape[0] * y.shape[2]
    print(f"B={B:4d} {secs:3d} s  {ms * 1e3:9.1f} us  {nbytes / ms / 1e6:8.1f} GB/s  {frames / ms / 1e3:8.1f} Mframes/s  "
          f"{26e3 * frames / ms / 1e9:6.1f} TFLOP/s(est)", flush=True)
