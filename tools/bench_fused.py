"""Micro-benchmark: activation + convolution of an AMP block as two kernels (dmel_aa_snake_f32, dmel_conv_forward at the fp16 split)
and as the fused kernel (dmel_conv_snake_forward), on the vocoder shapes of the bench workload (GPU only).

    python tools/bench_fused.py [--iters 20] [--only NAME]
Interleaved rounds in one process (the guide's rule 24); prints microseconds per launch and algorithmic TFLOP/s of the convolution."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib  # noqa: E402

SHAPES = {
    # name: (C, k, dil, T, B)
    "bv1_k3": (256, 3, 1, 736, 32), "bv1_k3d5": (256, 3, 5, 736, 32), "bv1_k7": (256, 7, 3, 736, 32), "bv1_k11": (256, 11, 5, 736, 32), "bv1_k11d1": (256, 11, 1, 736, 32),
    "bv2_k3": (128, 3, 1, 5888, 32), "bv2_k7": (128, 7, 3, 5888, 32), "bv2_k7d1": (128, 7, 1, 5888, 32), "bv2_k11": (128, 11, 5, 5888, 32), "bv2_k11d1": (128, 11, 1, 5888, 32),
    "bv3_k3": (64, 3, 1, 11776, 32), "bv3_k7": (64, 7, 3, 11776, 32), "bv3_k11": (64, 11, 1, 11776, 32), "bv3_k11d5": (64, 11, 5, 11776, 32),
    "bv4_k3": (32, 3, 1, 23552, 32), "bv4_k7": (32, 7, 3, 23552, 32), "bv4_k11": (32, 11, 5, 23552, 32), "bv4_k11d1": (32, 11, 1, 23552, 32),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    L = _lib.lib()
    dev = torch.device("cuda:0")
    taps = torch.tensor([0.0020289647, 0.0093894657, -0.0255434588, -0.0576573834, 0.1285725832, 0.4432097971, 0.4432097971, 0.1285725832,
                         -0.0576573834, -0.0255434588, 0.0093894657, 0.0020289647], dtype=torch.float32)
    tot = [0.0, 0.0, 0.0]
    for name, (Cc, k, dil, T, B) in SHAPES.items():
        if args.only and args.only not in name:
            continue
        w = torch.randn(Cc, Cc, k) / (Cc * k) ** 0.5
        b = torch.randn(Cc)
        h = C.c_void_p()
        _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cc, Cc, k, dil))
        _lib.check(L.dmel_conv_set_precision(h, 3))
        x = torch.randn(B, Cc, T, device=dev)
        u = torch.empty_like(x)
        y = torch.empty_like(x)
        y2 = torch.empty_like(x)
        y3 = torch.empty_like(x)
        al, be = (torch.randn(Cc) * 0.3).to(dev), (torch.randn(Cc) * 0.3).to(dev)
        st = _lib.stream_ptr()

        def snake():
            _lib.check(L.dmel_aa_snake_f32(x.data_ptr(), u.data_ptr(), al.data_ptr(), be.data_ptr(), taps.data_ptr(), taps.data_ptr(), 1, B, Cc, T, st))

        def conv():
            _lib.check(L.dmel_conv_forward(h, u.data_ptr(), y.data_ptr(), B, T, st))

        def fused():
            _lib.check(L.dmel_conv_snake_forward(h, x.data_ptr(), None, y2.data_ptr(), al.data_ptr(), be.data_ptr(), taps.data_ptr(), taps.data_ptr(), 1, B, T, st))

        def lean():      # the producer / consumer workgroup WITHOUT the activation: a plain convolution of u
            _lib.check(L.dmel_conv_snake_forward(h, u.data_ptr(), None, y3.data_ptr(), None, None, None, None, 1, B, T, st))

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3

        for fn in (snake, conv, fused, lean):
            fn()
        best = [1e30, 1e30, 1e30, 1e30]
        for _ in range(args.rounds):
            for i, fn in enumerate((snake, conv, fused, lean)):
                best[i] = min(best[i], timed(fn))
        same = bool(torch.equal(y, y2)) and bool(torch.equal(y, y3))
        fl = 2.0 * B * T * Cc * Cc * k
        for i in range(3):
            tot[i] += best[i]
        print(f"{name:10s} C={Cc:4d} k={k:2d} d={dil} T={T:6d}x{B}: snake {best[0]:7.1f} us + conv {best[1]:7.1f} us = {best[0] + best[1]:7.1f} | fused {best[2]:7.1f} us "
              f"({fl / best[2] / 1e6:6.1f} TF/s, x{(best[0] + best[1]) / best[2]:.2f}) | lean conv {best[3]:7.1f} us (x{best[1] / best[3]:.2f} of conv) bit-identical {same}", flush=True)
        L.dmel_conv_destroy(h)
    print(f"sum: snake {tot[0]:.0f} + conv {tot[1]:.0f} = {tot[0] + tot[1]:.0f} us, fused {tot[2]:.0f} us")


if __name__ == "__main__":
    main()
