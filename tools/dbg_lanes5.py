import sys, os, copy, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import make_codec
import dmel_codec_amd.torch_ops
dev = torch.device("cuda:0")
if os.environ.get("DMEL_DBG_EXCL") == "1":
    from dmel_codec_amd import _lib as _l
    _l.check(_l.lib().dmel_stft_set_exclusive_cu(1), "excl")
    print("exclusive CU mode on", flush=True)
codec = make_codec(720, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3).to(dev)
codec2 = copy.deepcopy(codec)
gen = torch.Generator().manual_seed(5)
L = 24000
batches = [(0.3 * torch.randn(3, 1, L, generator=gen)).to(dev) for _ in range(4)]
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

def run(name, fB, trials=120, nA=6):
    ref = [codec.encode_mel_transform(a).clone() for a in batches]
    torch.cuda.synchronize()
    bad = 0
    for t in range(trials):
        cur = torch.cuda.current_stream()
        sA.wait_stream(cur); sB.wait_stream(cur)
        with torch.cuda.stream(sB):
            for rep in range(12):
                fB()
        with torch.cuda.stream(sA):
            outs = [codec.encode_mel_transform(batches[j % 4]) for j in range(nA * 4)]
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref[j % 4]) else 1 for j, o in enumerate(outs))
    print(f"stft || {name}: {bad} wrong mel tensors of {trials * nA * 4}", flush=True)

big = torch.randn(8, 256, 736, device=dev); wb = torch.randn(256, 256, 3, device=dev) * 0.05; bb = torch.zeros(256, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
if which == "conv":
    run("conv1d_dilated 256x256 k3 T=736 (NP=3)", lambda: torch.ops.dmel_hip.conv1d_dilated(big, wb, bb, 1))
elif which == "sin":
    z = torch.randn(64, 1024, 1024, device=dev)
    run("torch.sin on 64 M elements", lambda: torch.sin(z), trials=60)
elif which == "snake":
    from dmel_codec_amd.models.modules.bigvgan.alias_free_activation.act import Activation1d
    from dmel_codec_amd.models.modules.bigvgan.activations import SnakeBeta
    act = Activation1d(SnakeBeta(128, alpha_logscale=True)).to(dev)
    xs = torch.randn(8, 128, 5888, device=dev)
    run("aa_snake 8 x 128 x 5888", lambda: act(xs))
elif which == "stft":
    au = torch.randn(64, 240000, device=dev) * 0.1
    run("stft 64 x 10 s", lambda: codec2.encode_mel_transform(au), trials=60)
elif which == "matmul":
    m1 = torch.randn(4096, 4096, device=dev)
    run("torch.matmul 4096^3 (rocBLAS)", lambda: m1 @ m1, trials=60)
elif which in ("poison_vgpr", "poison_lds"):
    import ctypes as C
    P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libpoison.so"))
    P.poison_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    sink = torch.zeros(256, device=dev)
    mode = 0 if which == "poison_vgpr" else 1
    run(which, lambda: P.poison_launch(mode, 4096, 200, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
