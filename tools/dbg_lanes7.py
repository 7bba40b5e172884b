import sys, os, copy, torch, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import make_codec
import dmel_codec_amd.torch_ops
from dmel_codec_amd import _lib
dev = torch.device("cuda:0")
if os.environ.get("DMEL_DBG_EXCL") == "1":
    from dmel_codec_amd import _lib as _l
    _l.check(_l.lib().dmel_stft_set_exclusive_cu(1), "excl")
    print("exclusive CU mode on", flush=True)
codec = make_codec(720, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3).to(dev)
gen = torch.Generator().manual_seed(5)
batches = [(0.3 * torch.randn(3, 1, 24000, generator=gen)).to(dev) for _ in range(4)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
big = torch.randn(8, 256, 736, device=dev)
w = torch.randn(256, 256, 3) * 0.05; b = torch.zeros(256)
L = _lib.lib()
def mk(prec):
    h = C.c_void_p(); _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), 256, 256, 3, 1)); _lib.check(L.dmel_conv_set_precision(h, prec)); return h
def conv(h):
    y = torch.empty(8, 256, 736, device=dev)
    _lib.check(L.dmel_conv_forward(h, big.data_ptr(), y.data_ptr(), 8, 736, _lib.stream_ptr()))
    return y
ref_mel = [codec.encode_mel_transform(a).clone() for a in batches]
for name, prec, env in (("NP=3 six-product", 0, {}), ("NP=2 fp16 split (conv_bf16)", 3, {"DMEL_CONV_PC": "0"}), ("NP=2 conv_pc", 3, {"DMEL_CONV_PC": "2"}), ("native fp32 MFMA", 2, {})):
    for k, v in env.items(): os.environ[k] = v
    h = mk(prec)
    ref_y = conv(h).clone(); torch.cuda.synchronize()
    bad_mel = bad_y = 0
    for t in range(100):
        cur = torch.cuda.current_stream(); sA.wait_stream(cur); sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            ys = [conv(h) for _ in range(12)]
        with torch.cuda.stream(sA):
            outs = [codec.encode_mel_transform(batches[j % 4]) for j in range(24)]
        torch.cuda.synchronize()
        bad_mel += sum(0 if torch.equal(o, ref_mel[j % 4]) else 1 for j, o in enumerate(outs))
        bad_y += sum(0 if torch.equal(y, ref_y) else 1 for y in ys)
    print(f"stft || conv {name}: wrong mel {bad_mel}/2400, wrong conv outputs {bad_y}/1200", flush=True)
    L.dmel_conv_destroy(h)
