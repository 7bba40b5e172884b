"""Micro-benchmark of the implicit-GEMM conv kernel on the shapes of the bench workload (GPU only).

    python tools/bench_conv.py [--iters 20] [--only NAME]
Prints algorithmic TFLOP/s per shape (hipEvent timing on the launch stream)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib  # noqa: E402

SHAPES = {
    # name: (Cout, Cin, k, dil, T, B)
    "bv1_k3": (256, 256, 3, 1, 736, 32), "bv1_k7": (256, 256, 7, 3, 736, 32), "bv1_k11": (256, 256, 11, 5, 736, 32),
    "bv2_k3": (128, 128, 3, 1, 5888, 32), "bv2_k7": (128, 128, 7, 3, 5888, 32), "bv2_k11": (128, 128, 11, 5, 5888, 32),
    "bv3_k7": (64, 64, 7, 3, 11776, 32), "bv3_k11": (64, 64, 11, 1, 11776, 32),
    "bv4_k3": (32, 32, 3, 1, 23552, 32), "bv4_k7": (32, 32, 7, 3, 23552, 32), "bv4_k11": (32, 32, 11, 5, 23552, 32),
    "wn_dec_1x1": (1120, 560, 1, 1, 92, 32), "wn_dec_k3": (1120, 560, 3, 2, 92, 32),
    "wn_enc_k3": (140, 70, 3, 4, 93, 256), "conv_pre": (512, 80, 7, 1, 92, 32),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--precision", type=int, default=0, help="0 fp32 (six-product bf16 split), 1 bf16 operands, 2 fp32 on the native fp32 MFMA, 3 fp32 from the three-product fp16 split")
    ap.add_argument("--check", action="store_true", help="compare with torch conv1d (operands rounded the same way)")
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    for name, (Cout, Cin, k, dil, T, B) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
        b = torch.randn(Cout)
        h = C.c_void_p()
        _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cout, Cin, k, dil))
        _lib.check(L.dmel_conv_set_precision(h, args.precision))
        x = torch.randn(B, Cin, T, device=dev)
        y = torch.empty(B, Cout, T, device=dev)
        st = _lib.stream_ptr()
        for _ in range(3):
            _lib.check(L.dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), B, T, st))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            _lib.check(L.dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), B, T, st))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 2.0 * B * T * Cout * Cin * k
        print(f"{name:12s} M={Cout:5d} K={Cin * k:5d} N={T}x{B}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
        if args.check:
            wd, xd = w.to(dev), x[:2]
            if args.precision == 1:
                wd, xd = wd.bfloat16().float(), xd.bfloat16().float()
            ref = torch.nn.functional.conv1d(xd.double(), wd.double(), b.to(dev).double(), dilation=dil, padding=dil * (k - 1) // 2)
            err = (y[:2].double() - ref).abs().max().item() / ref.abs().max().item()
            print(f"             max err / max |ref| = {err:.2e}", flush=True)
        L.dmel_conv_destroy(h)


if __name__ == "__main__":
    main()
