"""Micro-benchmark of the anti-aliased snake kernel on the BigVGAN-base stage shapes (GPU only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib
from oracle import ref_cpu

L = _lib.lib()
dev = torch.device("cuda:0")
taps = ref_cpu.aa_filter12().view(-1).contiguous()
for (B, C, T) in ((32, 256, 736), (32, 128, 5888), (32, 64, 11776), (32, 32, 23552)):
    x = torch.randn(B, C, T, device=dev)
    y = torch.empty_like(x)
    al, be = torch.randn(C, device=dev) * 0.3, torch.randn(C, device=dev) * 0.3
    st = _lib.stream_ptr()
    for _ in range(3):
        _lib.check(L.dmel_aa_snake_f32(x.data_ptr(), y.data_ptr(), al.data_ptr(), be.data_ptr(), taps.data_ptr(), taps.data_ptr(), 1, B, C, T, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(L.dmel_aa_snake_f32(x.data_ptr(), y.data_ptr(), al.data_ptr(), be.data_ptr(), taps.data_ptr(), taps.data_ptr(), 1, B, C, T, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"C={C:4d} T={T:6d}  {ms * 1e3:7.1f} us  {8.0 * B * C * T / ms / 1e6:7.1f} GB/s", flush=True)
