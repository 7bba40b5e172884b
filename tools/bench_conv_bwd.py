"""Micro-benchmark of conv backward-data / backward-weight on the bench workload's shapes (GPU only)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmel_codec_amd import _lib
from bench_conv import SHAPES

L = _lib.lib()
dev = torch.device("cuda:0")
for name, (Cout, Cin, k, dil, T, B) in SHAPES.items():
    w = torch.randn(Cout, Cin, k) / (Cin * k) ** 0.5
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), None, Cout, Cin, k, dil))
    x, dy = torch.randn(B, Cin, T, device=dev), torch.randn(B, Cout, T, device=dev)
    dx, dw = torch.empty_like(x), torch.empty(Cout, Cin, k, device=dev)
    st = _lib.stream_ptr()
    res = []
    for fn in (lambda: L.dmel_conv_backward_data(h, dy.data_ptr(), dx.data_ptr(), B, T, st),
               lambda: L.dmel_conv_backward_weight(h, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, B, T, st)):
        for _ in range(2):
            _lib.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.check(fn())
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * B * T * Cout * Cin * k
    print(f"{name:12s} dgrad {res[0] * 1e3:8.1f} us {fl / res[0] / 1e9:6.1f} TF/s   wgrad {res[1] * 1e3:8.1f} us {fl / res[1] / 1e9:6.1f} TF/s", flush=True)
    L.dmel_conv_destroy(h)
