"""GPU parity tests: the HIP path (through the C ABI of libdmel_hip.so) against the CPU oracle
(oracle/ref_cpu.py) on the same seeded inputs and against the golden fixtures generated from the reference.

Bars (BASELINE.json north_star): integer outputs (token ids) bit-exact; fp32 tensors within 1e-4 relative,
where relative = max|a-b| / max|b| (conftest.rel_err)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, report
from oracle import ref_cpu

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def cpu_sd(module):
    return {k: v.detach().cpu().float() for k, v in module.state_dict().items()}


def randomise(module, seed, scale=1.0):
    """O(1) weights so every term of the arithmetic matters (default inits are ~0.02 / 1e-6)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.split(".")[-1]
            if leaf in ("alpha", "beta"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
            elif leaf == "gamma":
                p.copy_(torch.randn(p.shape, generator=g) * 0.5)
            elif leaf == "weight_g":
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif p.ndim >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (scale / p[0].numel() ** 0.5))
            elif leaf == "weight":
                p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.2)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)


# ------------------------------------------------------------------------------------ STFT / log-mel
def assert_logmel_close(y, ref, what=""):
    """fp32 parity bar for log-mel: 1e-4 relative on the mel energies, with an absolute floor of 1e-6 of full
    scale (-120 dB).  Below that floor a tonal input's bins hold nothing but the fp32 rounding noise of the FFT
    itself (windowing alone injects ~6e-8 relative per sample), which differs between ANY two FFT implementations
    and which the log then magnifies; the oracle's torch.stft is no closer to an fp64 DFT there than the kernel is
    (test_stft_noise_floor_vs_fp64)."""
    a, b = torch.exp(y.detach().double().cpu()), torch.exp(ref.double())
    tol = 1e-4 * b + 1e-6 * b.max()
    bad = (a - b).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} mel values differ, worst {float(((a - b).abs() / tol).max()):.2f}x tolerance"


def test_stft_noise_floor_vs_fp64(dev, golden):
    """The sweep fixture has ~160 dB of dynamic range per frame; check the kernel against an fp64 DFT and show it
    is as close to the truth as the fp32 oracle is."""
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    g = golden("stft_logmel")
    x = g.ins["24k100"][:1]
    m = LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=100, f_min=0, f_max=12000)
    y = torch.exp(m(x.to(dev)).double().cpu())
    xp = F.pad(x.double().unsqueeze(1), (384, 384), mode="reflect")[0, 0]
    frames = xp.unfold(0, 1024, 256) * torch.hann_window(1024, dtype=torch.float64)
    mag = torch.sqrt(torch.fft.rfft(frames, dim=-1).abs() ** 2 + 1e-9).T
    truth = torch.clamp(torch.from_numpy(ref_cpu.slaney_mel_basis(24000, 1024, 100, 0.0, 12000.0)).double() @ mag, min=1e-5)
    oracle = torch.exp(g.outs["24k100"][0].double())
    err_gpu = ((y[0] - truth).abs() / truth).max()
    err_oracle = ((oracle - truth).abs() / truth).max()
    assert err_gpu < max(2.0 * err_oracle, 1e-5), (float(err_gpu), float(err_oracle))


def test_stft_logmel_golden(dev, golden):
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    g = golden("stft_logmel")
    for tag, (sr, n_mels, fmax) in g.meta["cases"].items():
        m = LogMelSpectrogram(sample_rate=sr, n_fft=1024, win_length=1024, hop_length=256, n_mels=n_mels, f_min=0, f_max=fmax)
        y = m(g.ins[tag].to(dev))
        assert y.shape == g.outs[tag].shape
        assert_logmel_close(y, g.outs[tag], tag)
        assert rel_err(y[1], g.outs[tag][1]) < TOL, tag       # the broadband (noise) clip: plain log-domain bar
        # the library's mel basis against the oracle's restatement
        assert torch.allclose(m.spectrogram.mel_basis(), torch.from_numpy(ref_cpu.slaney_mel_basis(sr, 1024, n_mels, 0.0, fmax)),
                              atol=1e-9)


@pytest.mark.parametrize("L,hop,B", [(385 + 640, 256, 1), (24000, 256, 3), (16000, 256, 2), (7777, 128, 2), (24000 * 10, 256, 2),
                                      (5000, 320, 1)])
def test_stft_logmel_random(dev, L, hop, B):
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    torch.manual_seed(L)
    x = torch.randn(B, 1, L) * 0.3
    m = LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=hop, n_mels=100, f_min=0, f_max=12000)
    y = m(x.to(dev))
    ref = ref_cpu.stft_logmel(x, 24000, 1024, 1024, hop, 100, 0.0, 12000.0)
    assert y.shape == ref.shape == (B, 100, 1 + (L + 2 * ((1024 - hop) // 2) - 1024) // hop)
    assert rel_err(y, ref) < TOL
    assert_logmel_close(y, ref)
    # (B, L) input and the fused length mask
    lens = torch.tensor([L - 300 * i for i in range(B)])
    y2 = m(x[:, 0].to(dev), lengths=lens.to(dev))
    mask = ref_cpu.sequence_mask(lens // hop, ref.shape[2])[:, None, :].float()
    assert rel_err(y2, ref * mask) < TOL
    assert torch.equal(y2.cpu() == 0, (ref * mask) == 0)


@pytest.mark.parametrize("sr,n_fft,hop,n_mels,win", [(44100, 2048, 512, 128, 2048), (24000, 512, 128, 80, 512),
                                                    (22050, 1024, 256, 80, 800), (44100, 2048, 441, 100, 1764)])
def test_stft_other_fft_sizes(dev, sr, n_fft, hop, n_mels, win):
    """n_fft 512 / 1024 / 2048 (the class default and the 44.1 kHz BigVGAN front end), win_length < n_fft, odd hops."""
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    torch.manual_seed(n_fft + hop)
    L = 30000 + hop // 3
    x = torch.randn(2, 1, L) * 0.2
    m = LogMelSpectrogram(sample_rate=sr, n_fft=n_fft, win_length=win, hop_length=hop, n_mels=n_mels, f_min=0, f_max=None)
    if (n_fft - hop) % 2:
        with pytest.raises(RuntimeError, match="hop_length"):
            m(x.to(dev))
        return
    y = m(x.to(dev))
    ref = ref_cpu.stft_logmel(x, sr, n_fft, win, hop, n_mels, 0.0, None)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < TOL
    assert_logmel_close(y, ref, f"n_fft={n_fft}")


def test_stft_rejects_unsupported(dev):
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    with pytest.raises(RuntimeError, match="n_fft"):
        LogMelSpectrogram(sample_rate=44100, n_fft=4096, win_length=4096, hop_length=1024, n_mels=128)(torch.zeros(1, 19000, device=dev))
    with pytest.raises(RuntimeError, match="GPU"):
        LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=80)(torch.zeros(1, 9000))
    with pytest.raises(RuntimeError, match="reflect pad"):
        LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=80)(torch.zeros(1, 300, device=dev))


# ------------------------------------------------------------------------------------ implicit-GEMM conv
@pytest.mark.parametrize("Cout,Cin,k,dil,T,B", [
    (32, 32, 3, 1, 100, 2), (32, 32, 11, 5, 300, 2), (64, 64, 7, 3, 257, 1), (128, 128, 3, 5, 130, 2),
    (256, 256, 11, 1, 96, 1), (140, 70, 3, 8, 93, 3), (1, 32, 7, 1, 500, 2), (512, 100, 7, 1, 92, 2),
    (70, 280, 1, 1, 23, 5), (100, 700, 1, 1, 92, 1), (33, 17, 5, 2, 65, 1),
])
def test_conv_igemm(dev, Cout, Cin, k, dil, T, B):
    from dmel_codec_amd import _lib
    torch.manual_seed(Cout * 1000 + Cin + k)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout) * 0.1
    x = torch.randn(B, Cin, T)
    ref = F.conv1d(x, w, b, dilation=dil, padding=dil * (k - 1) // 2)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cout, Cin, k, dil))
    xd = x.to(dev)
    y = torch.empty(B, Cout, T, device=dev)
    _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
    torch.cuda.synchronize()
    L.dmel_conv_destroy(h)
    assert rel_err(y, ref) < 2e-5


# ------------------------------------------------------------------------------------ anti-aliased snake
def test_activation1d_golden(dev, golden):
    from dmel_codec_amd.models.modules.bigvgan.alias_free_activation.act import Activation1d
    from dmel_codec_amd.models.modules.bigvgan import activations
    for kind in ("snakebeta", "snake"):
        g = golden("activation1d_" + kind)
        act = (activations.SnakeBeta if kind == "snakebeta" else activations.Snake)(6, alpha_logscale=True)
        m = Activation1d(activation=act)
        m.load_state_dict(g.sd)
        m = m.to(dev)
        for t in g.meta["lengths"]:
            y = m(g.ins[f"x{t}"].to(dev))
            assert rel_err(y, g.outs[f"y{t}"]) < 1e-5, (kind, t)


@pytest.mark.parametrize("B,Cc,T", [(2, 5, 1023), (1, 3, 1024), (2, 2, 1025), (1, 4, 5000), (3, 32, 333), (2, 3, 1017), (1, 2, 1018), (2, 2, 1019),
                                    (1, 3, 1024 + 6), (1, 2, 2036), (1, 2, 2037), (1, 3, 2044), (1, 2, 3)])
def test_activation1d_tiles(dev, B, Cc, T):
    from dmel_codec_amd import _lib
    torch.manual_seed(T)
    x = torch.randn(B, Cc, T) * 2
    alpha, beta = torch.randn(Cc) * 0.5, torch.randn(Cc) * 0.5
    for logscale in (True, False):
        a = alpha if logscale else alpha.abs() + 0.5
        bt = beta if logscale else beta.abs() + 0.5
        ref = ref_cpu.activation1d(x, a, bt, logscale=logscale)
        y = torch.empty(B, Cc, T, device=dev)
        taps = ref_cpu.aa_filter12().view(-1).contiguous()
        xd, ad, bd = x.to(dev), a.to(dev), bt.to(dev)      # keep the device buffers alive across the launch
        _lib.check(_lib.lib().dmel_aa_snake_f32(xd.data_ptr(), y.data_ptr(), ad.data_ptr(), bd.data_ptr(),
                                                taps.data_ptr(), taps.data_ptr(), int(logscale), B, Cc, T, _lib.stream_ptr()))
        assert rel_err(y, ref) < 1e-5


# ------------------------------------------------------------------------------------ WaveNet
def test_wavenet_golden(dev, golden):
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    g = golden("wavenet_enc")
    m = WaveNet(input_channels=10, residual_channels=70, residual_layers=g.meta["n_layers"], dilation_cycle=4)
    m.load_state_dict(g.sd)
    y = m.to(dev)(g.ins["x"].to(dev))
    assert rel_err(y, g.outs["y"]) < TOL
    g = golden("wavenet_dec")
    m = WaveNet(input_channels=48, output_channels=20, residual_channels=48, residual_layers=g.meta["n_layers"],
                dilation_cycle=4, condition_channels=48)
    m.load_state_dict(g.sd)
    y = m.to(dev)(g.ins["x"].to(dev), condition=g.ins["cond"].to(dev))
    assert rel_err(y, g.outs["y"]) < TOL


def test_wavenet_real_widths_and_masks(dev):
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    # encoder shape of the reference configs (10 -> 70, 20 layers), ragged lengths, group repeat
    m = WaveNet(input_channels=10, residual_channels=70, residual_layers=20, dilation_cycle=4)
    randomise(m, 1)
    torch.manual_seed(2)
    G, B, T = 4, 3, 93
    x = torch.randn(B * G, 10, T)
    lens = torch.tensor([93, 40, 77])
    mask = ref_cpu.sequence_mask(lens, T)[:, None, :].float().repeat_interleave(G, dim=0)
    ref = ref_cpu.wavenet_forward(cpu_sd(m), "", x, 20) * mask
    y = m.to(dev)(x.to(dev), out_lengths=lens.to(dev), group_repeat=G)
    assert rel_err(y, ref) < TOL
    # decoder shape (700 wide, conditioned, 100 out), 3 layers to keep the oracle quick
    m = WaveNet(input_channels=700, output_channels=100, residual_channels=700, residual_layers=3, dilation_cycle=4,
                condition_channels=700)
    randomise(m, 3)
    x, c = torch.randn(2, 700, 92), torch.randn(2, 700, 92)
    lens = torch.tensor([92, 60])
    mask = ref_cpu.sequence_mask(lens, 92)[:, None, :].float()
    ref = ref_cpu.wavenet_forward(cpu_sd(m), "", x * mask, 3, condition=c) * mask
    y = m.to(dev)(x.to(dev), condition=c.to(dev), in_lengths=lens.to(dev), out_lengths=lens.to(dev))
    assert rel_err(y, ref) < TOL


# ------------------------------------------------------------------------------------ quantiser
def near_tie_report(ids_gpu, ids_ref, pre_ref, eps=2e-4):
    """ids must match wherever the oracle's pre-round value is farther than eps from a rounding boundary
    (x.5); at a near-tie either neighbour is accepted.  pre_ref: (G, B, L, D); ids: (B, G, L)."""
    frac = (pre_ref - torch.floor(pre_ref) - 0.5).abs().amin(dim=-1)        # (G, B, L)
    tie = (frac < eps).permute(1, 0, 2)
    diff = ids_gpu.cpu() != ids_ref
    return int(diff.sum()), int((diff & ~tie).sum()), int(tie.sum())


def test_producer_consumer_convolution_is_bit_identical(dev, monkeypatch):
    """conv_pc.hip (eight consumer waves that only run the MFMA loop + four producer waves that stage x) serves the fp16-split launches of
    >= 5 row tiles -- the decoder WaveNet's gate (two segments, sigmoid * tanh epilogue) and residual / skip convolutions, the 256-channel
    vocoder stage -- unless DMEL_CONV_PC=0.  Same operand bits, K order and epilogue: torch.equal against conv_bf16_kernel on single
    convolutions (ragged channel counts, both halos, residual), on the conditioned WaveNet (with masks) and on a whole vocoder."""
    from dmel_codec_amd import _lib
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    L = _lib.lib()

    def both(fn):
        monkeypatch.setenv("DMEL_CONV_PC", "2")      # 2: also launches of fewer than 128 workgroups (the default leaves those to conv_bf16_kernel)
        a = fn()
        monkeypatch.setenv("DMEL_CONV_PC", "0")
        b = fn()
        return a, b

    for (Co, Ci, k, dil, T, B) in ((256, 256, 3, 1, 300, 2), (256, 256, 11, 5, 97, 1), (200, 168, 7, 3, 96, 2), (512, 80, 7, 1, 92, 3), (1120, 560, 3, 8, 92, 2)):
        torch.manual_seed(Co + k)
        w = torch.randn(Co, Ci, k) / math.sqrt(Ci * k)
        bb = torch.randn(Co) * 0.1
        x = torch.randn(B, Ci, T).to(dev)
        h = C.c_void_p()
        _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), bb.data_ptr(), Co, Ci, k, dil))
        _lib.check(L.dmel_conv_set_precision(h, 3))

        def run():
            y = torch.full((B, Co, T), float("nan"), device=dev)
            _lib.check(L.dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
            torch.cuda.synchronize()
            return y
        y1, y0 = both(run)
        assert torch.equal(y1, y0), (Co, Ci, k, dil, T, float((y1 - y0).abs().max()))
        assert rel_err(y1, F.conv1d(x.cpu().double(), w.double(), bb.double(), dilation=dil, padding=dil * (k - 1) // 2)) < 2e-5
        L.dmel_conv_destroy(h)
    # the conditioned decoder WaveNet: gate (x k3 dilated + condition 1x1, GATE epilogue) and residual / skip (1x1, RESSKIP epilogue)
    torch.manual_seed(23)
    m = WaveNet(input_channels=560, output_channels=80, residual_channels=560, residual_layers=5, dilation_cycle=4, condition_channels=560)
    randomise(m, 24)
    m.set_precision("fp32_f16x2")
    m = m.to(dev)
    g = torch.Generator().manual_seed(25)
    for (N, T, lens) in ((3, 92, None), (2, 200, [200, 77])):
        x, c = torch.randn(N, 560, T, generator=g).to(dev), torch.randn(N, 560, T, generator=g).to(dev)
        kw = {} if lens is None else dict(in_lengths=torch.tensor(lens).to(dev), out_lengths=torch.tensor(lens).to(dev))
        y1, y0 = both(lambda: m(x, condition=c, **kw))
        assert torch.equal(y1, y0), (N, T, float((y1 - y0).abs().max()))
    # a vocoder with a 256-channel first stage
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.configs import bigvgan_h
    hcfg = bigvgan_h("base_24k_100band", num_mels=80)
    torch.manual_seed(26)
    v = BigVGAN(hcfg)
    randomise(v, 27, scale=0.7)
    v = v.to(dev)
    mel = torch.randn(2, 80, 20, generator=g).to(dev)
    y1, y0 = both(lambda: v(mel))
    assert torch.equal(y1, y0)


@pytest.mark.parametrize("N,T,L,lens", [(3, 92, 4, None), (2, 300, 5, None), (4, 92, 3, [92, 50, 0, 91]), (1, 937, 2, None)])
def test_decoder_wavenet_with_presplit_operands_is_bit_identical(dev, monkeypatch, N, T, L, lens):
    """The conditioned 560-channel decoder WaveNet at the fp16-split precision reads every convolution input as pre-split fp16 planes
    (condition split once per forward, gate output written by its epilogue as planes, residual stream as fp32 + planes) instead of
    converting in the staging pass once per layer and row block.  Same operand bits -> the output must be torch.equal to the path that
    converts in the staging pass (DMEL_WAVENET_PRESPLIT=0), on the 128 x 96 and the eight-wave 256 x 96 tiles, with length masks; and
    within the parity bar of the oracle."""
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    torch.manual_seed(17 + T)
    m = WaveNet(input_channels=560, output_channels=80, residual_channels=560, residual_layers=L, dilation_cycle=4, condition_channels=560)
    randomise(m, 18)
    m.set_precision("fp32_f16x2")
    g = torch.Generator().manual_seed(19)
    x, c = torch.randn(N, 560, T, generator=g), torch.randn(N, 560, T, generator=g)
    ln = None if lens is None else torch.tensor(lens)
    sd = cpu_sd(m)
    m = m.to(dev)
    kw = {} if ln is None else dict(in_lengths=ln.to(dev), out_lengths=ln.to(dev))
    monkeypatch.setenv("DMEL_WAVENET_PRESPLIT", "1")
    y1 = m(x.to(dev), condition=c.to(dev), **kw)
    monkeypatch.setenv("DMEL_WAVENET_PRESPLIT", "0")
    y0 = m(x.to(dev), condition=c.to(dev), **kw)
    assert torch.equal(y1, y0), float((y1 - y0).abs().max())
    if ln is None:
        ref = ref_cpu.wavenet_forward(sd, "", x, L, condition=c)
        assert rel_err(y1, ref) < TOL


@pytest.mark.parametrize("levels,prebound", [([7, 5, 5], True), ([7, 5, 5], False), ([8, 6], True)])
def test_quantizer_encode_decode(dev, levels, prebound):
    from dmel_codec_amd.models.modules.dowmsample_fsq import DownsampleFiniteScalarQuantize
    G, Cg, B, T = 10, 70, 3, 93
    q = DownsampleFiniteScalarQuantize(input_dim=G * Cg, n_codebooks=1, n_groups=G, levels=levels, downsample_factor=(2, 2),
                                       is_dmel=True, fsq_prebound=prebound)
    randomise(q, 5, scale=1.5)
    torch.manual_seed(6)
    z = torch.randn(B * G, Cg, T)
    sd = cpu_sd(q)
    ids_ref, pre_ref = ref_cpu.quantizer_encode(sd, "", z, G, levels, (2, 2), prebound, return_prequant=True)
    ids, pre = q.to(dev).encode(z.to(dev), return_prequant=True)
    assert ids.dtype == torch.int32 and ids.shape == ids_ref.shape == (B, G, T // 4)
    assert (pre.cpu() - pre_ref).abs().max() < 5e-5
    n_diff, n_bad, n_tie = near_tie_report(ids, ids_ref, pre_ref)
    assert n_bad == 0, f"{n_bad} id mismatches away from rounding boundaries ({n_diff} total, {n_tie} near-ties)"
    assert len(torch.unique(ids_ref)) > 20          # the test exercises many codes
    # decode: table lookup + Linear + upsampling, compared on the ORACLE's ids
    zq_ref = ref_cpu.quantizer_decode(sd, "", ids_ref, G, levels, (2, 2))
    zq = q.decode(ids_ref.to(dev))
    assert zq.shape == zq_ref.shape == (B, G * Cg, (T // 4) * 4)
    assert rel_err(zq, zq_ref) < TOL


# ------------------------------------------------------------------------------------ activation fused into the convolution
@pytest.mark.parametrize("C_,k,dil,T,B,kind", [
    (32, 3, 1, 100, 2, "snakebeta"), (32, 11, 5, 1000, 2, "snakebeta"), (32, 7, 1, 384, 1, "snake"), (32, 7, 3, 385, 1, "snakebeta"),
    (64, 7, 3, 257, 1, "snakebeta"), (64, 3, 5, 192, 2, "snake"), (64, 11, 1, 600, 1, "snakebeta"),
    (128, 3, 5, 130, 2, "snakebeta"), (128, 11, 5, 96, 2, "snakebeta"), (128, 7, 1, 97, 1, "snake"), (128, 11, 3, 311, 1, "snakebeta"),
    (256, 11, 1, 96, 1, "snakebeta"), (256, 3, 3, 200, 2, "snake"), (256, 7, 5, 95, 1, "snakebeta"),
    (48, 7, 1, 64, 2, "snakebeta"), (96, 3, 1, 7, 1, "snakebeta"), (24, 11, 5, 1, 2, "snakebeta"), (200, 3, 1, 50, 1, "snake"),
    (384, 3, 1, 120, 1, "snakebeta"), (16, 7, 5, 2, 1, "snakebeta"), (128, 3, 1, 13, 1, "snakebeta"),
])
def test_conv_with_fused_activation_is_bit_identical_to_the_two_kernel_path(dev, C_, k, dil, T, B, kind):
    """dmel_conv_snake_forward (act -> conv of AMPBlock1/2.forward, bigvgan.py:132-141, as one kernel with producer / consumer waves)
    against dmel_aa_snake_f32 + dmel_conv_forward at the same precision: torch.equal, with and without the residual; and against the
    oracle (Activation1d restated + F.conv1d, float64) at the conv bar.  Shapes: every consumer layout (32 / 64 / 128 / 256 rows, ragged
    channel counts, more rows than one workgroup holds), both halos, tiles that end inside a row, rows shorter than the filters."""
    from dmel_codec_amd import _lib
    torch.manual_seed(C_ * 131 + k * 7 + dil + T)
    w = torch.randn(C_, C_, k) / math.sqrt(C_ * k)
    b = torch.randn(C_) * 0.1
    x = torch.randn(B, C_, T) * 1.5
    res = torch.randn(B, C_, T)
    alpha, beta = torch.randn(C_) * 0.5, torch.randn(C_) * 0.5
    taps = ref_cpu.aa_filter12().view(-1).contiguous()
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), C_, C_, k, dil))
    _lib.check(L.dmel_conv_set_precision(h, 3))                     # DMEL_PRECISION_FP32_F16X2: what the fused kernel computes in
    xd, rd, ad = x.to(dev), res.to(dev), alpha.to(dev)
    bd = beta.to(dev) if kind == "snakebeta" else None
    for logscale in (1, 0):
        a_ = ad if logscale else ad.abs() + 0.5
        b_ = None if bd is None else (bd if logscale else bd.abs() + 0.5)
        u = torch.empty_like(xd)
        _lib.check(L.dmel_aa_snake_f32(xd.data_ptr(), u.data_ptr(), a_.data_ptr(), _lib.ptr(b_), taps.data_ptr(), taps.data_ptr(), logscale,
                                       B, C_, T, _lib.stream_ptr()))
        y2 = torch.empty(B, C_, T, device=dev)
        _lib.check(L.dmel_conv_forward(h, u.data_ptr(), y2.data_ptr(), B, T, _lib.stream_ptr()))
        y1 = torch.full((B, C_, T), float("nan"), device=dev)
        _lib.check(L.dmel_conv_snake_forward(h, xd.data_ptr(), None, y1.data_ptr(), a_.data_ptr(), _lib.ptr(b_), taps.data_ptr(),
                                             taps.data_ptr(), logscale, B, T, _lib.stream_ptr()))
        torch.cuda.synchronize()
        nbad = int((y1 != y2).sum())
        assert nbad == 0, f"fused != two kernels at {nbad} of {y1.numel()} outputs (max diff {float((y1 - y2).abs().max()):.3e}), logscale {logscale}"
        y1r = torch.full((B, C_, T), float("nan"), device=dev)
        _lib.check(L.dmel_conv_snake_forward(h, xd.data_ptr(), rd.data_ptr(), y1r.data_ptr(), a_.data_ptr(), _lib.ptr(b_), taps.data_ptr(),
                                             taps.data_ptr(), logscale, B, T, _lib.stream_ptr()))
        assert torch.equal(y1r, y2 + rd)
        a_c = a_.cpu().double()
        b_c = a_c if b_ is None else b_.cpu().double()
        f64 = ref_cpu.aa_filter12().double()
        ref = F.conv1d(ref_cpu.activation1d(x.double(), a_c, b_c, f64, f64, logscale=bool(logscale)), w.double(), b.double(),
                       dilation=dil, padding=dil * (k - 1) // 2)
        assert rel_err(y1, ref) < 2e-5
    L.dmel_conv_destroy(h)


def test_bigvgan_fused_and_two_kernel_paths_agree_bit_for_bit(dev, golden, monkeypatch):
    """The whole vocoder with the activations fused into the convolutions (DMEL_FUSE_SNAKE=1) and as activation kernel + convolution
    kernel (the default): same bits, one stream or three, AMPBlock1 and AMPBlock2."""
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.models.modules.bigvgan.env import AttrDict
    for name in ("bigvgan_tiny", "bigvgan_tiny_ampblock2"):
        g = golden(name)
        m = BigVGAN(AttrDict(dict(g.meta["h"])))
        m.load_state_dict(g.sd)
        m = m.to(dev)
        mel = g.ins["mel"].to(dev)
        monkeypatch.setenv("DMEL_FUSE_SNAKE", "1")
        y_f = m(mel)
        monkeypatch.setenv("DMEL_FUSE_SNAKE", "0")
        y_u = m(mel)
        assert torch.equal(y_f, y_u), (name, float((y_f - y_u).abs().max()))
        m.set_streams(1)
        monkeypatch.setenv("DMEL_FUSE_SNAKE", "1")
        assert torch.equal(m(mel), y_f)
        assert rel_err(y_f, g.outs["audio"]) < TOL


# ------------------------------------------------------------------------------------ BigVGAN
def test_bigvgan_golden(dev, golden):
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.models.modules.bigvgan.env import AttrDict
    g = golden("bigvgan_tiny")
    m = BigVGAN(AttrDict(dict(g.meta["h"])))
    m.load_state_dict(g.sd)
    y = m.to(dev)(g.ins["mel"].to(dev))
    assert y.shape == g.outs["audio"].shape
    assert rel_err(y, g.outs["audio"]) < TOL
    g = golden("bigvgan_tiny_snake_nowm")
    m = BigVGAN(AttrDict(dict(g.meta["h"])))
    m.remove_weight_norm()
    m.load_state_dict(g.sd)
    y = m.to(dev)(g.ins["mel"].to(dev))
    assert rel_err(y, g.outs["audio"]) < TOL


def test_bigvgan_ampblock2_golden(dev, golden):
    """`resblock: "2"` (AMPBlock2, bigvgan.py:150-241): the HIP path against outputs of the reference's own class."""
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.models.modules.bigvgan.env import AttrDict
    g = golden("bigvgan_tiny_ampblock2")
    m = BigVGAN(AttrDict(dict(g.meta["h"])))
    m.load_state_dict(g.sd)
    y = m.to(dev)(g.ins["mel"].to(dev))
    assert y.shape == g.outs["audio"].shape
    assert rel_err(y, g.outs["audio"]) < TOL
    m.set_streams(1)
    assert torch.equal(m(g.ins["mel"].to(dev)), y)           # one stream or three: same bits
    m.remove_weight_norm()
    assert rel_err(m(g.ins["mel"].to(dev)), g.outs["audio"]) < TOL


def assert_close_to_truth(y, ref32, ref64, what=""):
    """Deep random-weight stacks amplify rounding noise: the fp32 oracle itself moves by 3-5e-5 relative when only
    its thread count (oneDNN blocking) changes.  So whole-network outputs are judged against the SAME oracle run in
    float64: the kernel must be within 1e-4 of the truth or within 1.5x the fp32 oracle's own distance from it.  Every call puts its
    three numbers on record (gpurun_out/parity_report.txt -> profiles/): how much of the allowance is used is visible, not assumed."""
    e_gpu, e_ref, e_32 = rel_err(y, ref64), rel_err(ref32, ref64), rel_err(y, ref32)
    report(f"[truth] {what}: gpu-vs-fp64 {e_gpu:.2e}, oracle-fp32-vs-fp64 {e_ref:.2e}, gpu-vs-oracle-fp32 {e_32:.2e}")
    # round 3: every recorded case has e_gpu < e_ref (profiles/r03_parity_report.txt), so the relief factor went from 3 to 1.5
    assert e_gpu < max(TOL, 1.5 * e_ref), f"{what}: gpu-vs-fp64 {e_gpu:.2e}, oracle-fp32-vs-fp64 {e_ref:.2e}"
    # (round 2 also allowed the two fp32 results 4x max(TOL, e_ref) of each other: that bound follows from the one above by the triangle
    # inequality -- e_32 <= e_gpu + e_ref -- and is no longer asserted separately; e_32 is on record instead)


def to64(sd):
    return {k: v.double() for k, v in sd.items()}


def test_bigvgan_base_config(dev):
    """BigVGAN-base (BASELINE config 2 vocoder, 14 M parameters) on a short mel against the oracle."""
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.configs import bigvgan_h
    h = bigvgan_h("base_24k_100band", num_mels=80)
    torch.manual_seed(11)
    m = BigVGAN(h)
    randomise(m, 12, scale=0.7)
    mel = torch.randn(2, 80, 12)
    sd = cpu_sd(m)
    ref = ref_cpu.bigvgan_forward(sd, dict(h), mel)
    ref64 = ref_cpu.bigvgan_forward(to64(sd), dict(h), mel.double())
    y = m.to(dev)(mel.to(dev))
    assert y.shape == ref.shape == (2, 1, 12 * 256)
    assert_close_to_truth(y, ref, ref64, "bigvgan-base")


@pytest.mark.parametrize("name", ["v2_24k_100band_256x", "v2_44k_128band_512x"])
def test_bigvgan_v2_configs(dev, name):
    """The 112 M / 122 M parameter v2 vocoders (BASELINE config 4; six stages, 1536 -> 24 channels, no final tanh/bias)
    on a 3-frame mel: exercises channel counts that are not multiples of 32 and every stage's transposed conv."""
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.configs import bigvgan_h
    h = bigvgan_h(name)
    torch.manual_seed(21)
    m = BigVGAN(h)
    randomise(m, 22, scale=0.7)
    with torch.no_grad():
        m.conv_post.weight_g.fill_(0.02)     # keep the random net's output inside the final clamp(-1, 1)
    mel = torch.randn(1, h.num_mels, 3)
    sd = cpu_sd(m)
    ref = ref_cpu.bigvgan_forward(sd, dict(h), mel)
    ref64 = ref_cpu.bigvgan_forward(to64(sd), dict(h), mel.double())
    y = m.to(dev)(mel.to(dev))
    up = math.prod(h.upsample_rates)
    assert y.shape == ref.shape == (1, 1, 3 * up)
    assert float(ref.abs().max()) < 0.999 and float(ref.abs().mean()) > 1e-3      # neither saturated nor dead
    assert_close_to_truth(y, ref, ref64, name)


# ------------------------------------------------------------------------------------ whole codec
def make_codec(seed, **kw):
    from dmel_codec_amd.configs import build_codec
    torch.manual_seed(seed)
    codec = build_codec(**kw)
    randomise(codec.encoder, seed + 1)
    randomise(codec.quantizer, seed + 2, scale=1.5)
    randomise(codec.decoder, seed + 3)
    if codec.vocoder is not None:
        randomise(codec.vocoder, seed + 4, scale=0.7)
    with torch.no_grad():
        codec.quality_projection.weight.normal_(0, 0.3)
        codec.quality_projection.bias.normal_(0, 0.1)
    return codec


def split_sd(codec):
    sd = cpu_sd(codec)
    voc = {k[len("vocoder."):]: v for k, v in sd.items() if k.startswith("vocoder.")}
    return {k: v for k, v in sd.items() if not k.startswith("vocoder.")}, voc


@pytest.mark.parametrize("n_mels,G,sr,L", [(80, 8, 16000, 16000), (100, 10, 24000, 24000)])
def test_codec_encode_ids_bit_exact(dev, n_mels, G, sr, L):
    """BASELINE config 1 (1 s, 16 kHz, 80 mel, 8 groups) and the reference default (24 kHz, 100 mel, 10 groups):
    encode() token ids against the oracle, ragged batch."""
    from dmel_codec_amd.configs import oracle_cfg
    codec = make_codec(100 + G, sample_rate=sr, n_mels=n_mels, dmel_groups=G, vocoder=None,
                       f_max=None if sr == 16000 else 12000.0)
    cfg = oracle_cfg(codec)
    sd, _ = split_sd(codec)
    gen = torch.Generator().manual_seed(1234)
    audio = torch.randn(3, 1, L, generator=gen)
    audio = 0.95 * audio / audio.abs().amax(dim=-1, keepdim=True)
    lens = torch.tensor([[L, L - 5000, L // 2]])            # (1, B) as the reference's collate emits
    ids_ref, lens_ref, pre_ref = ref_cpu.vqgan_encode(sd, cfg, audio, lens, return_prequant=True)
    codec = codec.to(dev)
    ids, ilens = codec.encode(audio.to(dev), lens.to(dev))
    assert ids.dtype == torch.int32 and ids.shape == ids_ref.shape == (3, G, (L // 256) // 4)
    assert torch.equal(ilens.cpu(), lens_ref)
    n_diff, n_bad, n_tie = near_tie_report(ids, ids_ref, pre_ref)
    report(f"[ids] encode {n_mels} mel / {G} groups / {sr} Hz, 3 ragged clips: n_diff={n_diff} n_tie(2e-4)={n_tie} of {ids.numel()} ids")
    assert n_bad == 0, f"{n_bad} id mismatches away from rounding boundaries ({n_diff} total, {n_tie} near-ties)"
    assert n_diff <= n_tie
    assert int(ids.max()) < math.prod(cfg["levels"]) and int(ids.min()) >= 0


def test_codec_decode_waveform(dev):
    """decode(): ids -> quantised latent -> conditional WaveNet (noise injected) -> BigVGAN, against the oracle."""
    from dmel_codec_amd.configs import oracle_cfg
    codec = make_codec(300, n_mels=80, dmel_groups=8, decoder_layers=4, encoder_layers=2)
    cfg = oracle_cfg(codec)
    sd, voc_sd = split_sd(codec)
    B, T4 = 2, 6
    gen = torch.Generator().manual_seed(4321)
    ids = torch.randint(0, 175, (B, 8, T4), generator=gen, dtype=torch.int32)
    flen = torch.tensor([T4, 4])
    noise = torch.randn(B, 560, T4 * 4, generator=gen)
    audio_ref, mel_ref = ref_cpu.vqgan_decode(sd, cfg, ids, flen, noise, voc_sd, dict(codec.vocoder.h))
    codec = codec.to(dev)
    audio, mel = codec.decode(ids.to(dev), flen.to(dev), return_audios=True, noise=noise.to(dev))
    assert mel.shape == mel_ref.shape and audio.shape == audio_ref.shape == (B, 1, T4 * 4 * 256)
    assert rel_err(mel, mel_ref) < TOL
    audio64, _ = ref_cpu.vqgan_decode(to64(sd), cfg, ids, flen, noise.double(), to64(voc_sd), dict(codec.vocoder.h))
    assert_close_to_truth(audio, audio_ref, audio64, "decode waveform")
    z_ref, _ = ref_cpu.vqgan_quantized_features(sd, cfg, ids, flen)
    z, _ = codec.get_quantized_features_from_indices(ids.to(dev), flen.to(dev))
    assert rel_err(z, z_ref) < TOL


def test_codec_round_trip_properties(dev):
    """Size-independent properties at a larger size than the oracle is asked to check: determinism, batch
    independence (an item's ids do not depend on its neighbours) and mask behaviour."""
    codec = make_codec(500, n_mels=80, dmel_groups=8, vocoder=None).to(dev)
    gen = torch.Generator().manual_seed(7)
    B, L = 8, 24000 * 2
    audio = (torch.randn(B, 1, L, generator=gen) * 0.2).to(dev)
    lens = torch.full((B,), L, device=dev)
    ids1, l1 = codec.encode(audio, lens)
    ids2, _ = codec.encode(audio, lens)
    assert torch.equal(ids1, ids2)
    ids_sub, _ = codec.encode(audio[2:5], lens[2:5])
    assert torch.equal(ids_sub, ids1[2:5])
    assert int(l1[0]) == (L // 256) // 4
    noise = torch.randn(B, 560, ids1.shape[2] * 4, generator=gen).to(dev)
    mel = codec.decode(ids1, l1, noise=noise)
    short = l1.clone()
    short[0] = 5
    mel2 = codec.decode(ids1, short, noise=noise)
    assert torch.all(mel2[0, :, 20:] == 0) and torch.equal(mel2[1:], mel[1:])


def test_streaming_decode_is_bit_identical(dev):
    """decode_chunked (windows re-run with a halo and cropped) against decode() on the whole sequence: every layer has finite
    support, so with a halo >= the receptive field the concatenation must be bit-identical (12.8 s, ragged batch)."""
    codec = make_codec(700, n_mels=80, dmel_groups=8, encoder_layers=2).to(dev)
    gen = torch.Generator().manual_seed(9)
    B, T4 = 2, 300
    ids = torch.randint(0, 175, (B, 8, T4), generator=gen, dtype=torch.int32).to(dev)
    flen = torch.tensor([T4, 211], device=dev)
    noise = torch.randn(B, 560, T4 * 4, generator=gen).to(dev)
    audio, mel = codec.decode(ids, flen, return_audios=True, noise=noise)
    audio_c, mel_c = codec.decode_chunked(ids, flen, chunk_tokens=64, noise=noise)
    assert audio_c.shape == audio.shape == (B, 1, T4 * 4 * 256) and mel_c.shape == mel.shape
    assert torch.equal(mel_c, mel)
    assert torch.equal(audio_c, audio)
    # too small a halo must NOT be exact (the test would otherwise be vacuous)
    audio_bad, _ = codec.decode_chunked(ids, flen, chunk_tokens=64, halo_tokens=2, noise=noise)
    assert not torch.equal(audio_bad, audio)


@pytest.mark.parametrize("pattern", ["64", "ragged", "one_by_one"])
def test_incremental_decode_with_state_carry(dev, pattern):
    """decode_stream / StreamingDecoder: token chunks go in as an LM would emit them, audio comes out as soon as its right context
    exists, and the concatenation is BIT-identical to decode() on the whole sequence (ragged batch, injected noise).  The decoder
    WaveNet carries the output history of every block (dmel_wavenet_stream_step) instead of re-running halos, so the convolution work
    of the whole stream stays within 1.2x of the one-shot decode at 64-token chunks (the windowed form costs ~2x)."""
    from dmel_codec_amd import _lib
    codec = make_codec(700, n_mels=80, dmel_groups=8, encoder_layers=2).to(dev)
    gen = torch.Generator().manual_seed(9)
    B, T4 = 2, 300
    ids = torch.randint(0, 175, (B, 8, T4), generator=gen, dtype=torch.int32).to(dev)
    flen = torch.tensor([T4, 211], device=dev)
    noise = torch.randn(B, 560, T4 * 4, generator=gen).to(dev)
    _lib.prof_reset(); _lib.prof_enable(True)
    audio, mel = codec.decode(ids, flen, return_audios=True, noise=noise)
    torch.cuda.synchronize()
    whole = _lib.prof_read("conv_igemm")["flops"]
    _lib.prof_reset()
    if pattern == "64":
        sizes = [64] * 4 + [44]
    elif pattern == "ragged":
        sizes = [1, 7, 30, 100, 3, 120, 39]
    else:
        sizes = [1] * 40 + [260]
    assert sum(sizes) == T4
    dec = codec.streaming_decoder(B, flen, return_audios=True)
    pieces, pos, first_audio_after = [], 0, None
    for n in sizes:
        a, m = dec.push(ids[:, :, pos:pos + n], noise=noise[:, :, pos * 4:(pos + n) * 4])
        pos += n
        assert a.shape[-1] == m.shape[-1] * 256
        if m.shape[-1] and first_audio_after is None:
            first_audio_after = pos
        pieces.append((a, m))
    pieces.append(dec.finish())
    torch.cuda.synchronize()
    streamed = _lib.prof_read("conv_igemm")["flops"]
    _lib.prof_enable(False); _lib.prof_reset()
    audio_s = torch.cat([p[0] for p in pieces], dim=-1)
    mel_s = torch.cat([p[1] for p in pieces], dim=-1)
    assert mel_s.shape == mel.shape and audio_s.shape == audio.shape
    assert torch.equal(mel_s, mel)
    assert torch.equal(audio_s, audio)
    # latency in tokens: the first audio appears once WaveNet (75 frames) + vocoder (~20) + quantiser (16) context has arrived
    assert first_audio_after is not None and first_audio_after <= 64
    if pattern == "64":
        report(f"[stream] conv flops streamed / whole = {streamed / whole:.3f}, first audio after {first_audio_after} tokens")
        assert streamed <= 1.2 * whole, streamed / whole
    # the generator form over an arbitrary iterable of chunks
    parts = list(codec.decode_stream(iter([ids[:, :, :150], ids[:, :, 150:]]), flen, noise=None, return_audios=False))
    assert sum(p[1].shape[-1] for p in parts) == T4 * 4 and all(p[0] is None for p in parts)
    mel_g = torch.cat([p[1] for p in codec.decode_stream(ids, flen, chunk_tokens=50, noise=noise, return_audios=False)], dim=-1)
    assert torch.equal(mel_g, mel)
    # pipeline=True: the vocoder of chunk i on its own stream under the WaveNet of chunk i + 1, pieces one chunk later -- same pieces, same bits
    if pattern == "64":
        plain = list(codec.decode_stream(ids, flen, chunk_tokens=40, noise=noise))
        piped = list(codec.decode_stream(ids, flen, chunk_tokens=40, noise=noise, pipeline=True))
        assert len(plain) == len(piped)
        for (a0, m0), (a1, m1) in zip(plain, piped):
            assert torch.equal(a0, a1) and torch.equal(m0, m1)
        assert torch.equal(torch.cat([p[0] for p in piped], dim=-1), audio)
        with pytest.raises(ValueError, match="exclude"):
            codec.streaming_decoder(B, None, True, graph_chunk_tokens=32, overlap_vocoder=True)


def test_lanes_match_sequential(dev):
    """dmel_codec_amd.pipeline.CodecLanes: sixteen batches dealt to two lanes (replicas on their own streams, in flight together) give the
    ids and waveforms of the same batches through the single codec one after the other, bit for bit -- same kernels, same weights,
    and the decoder's Gaussian input is drawn in submission order."""
    from dmel_codec_amd.pipeline import CodecLanes
    codec = make_codec(720, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3).to(dev)
    gen = torch.Generator().manual_seed(5)
    L = 24000
    NB = 16      # enough batches in flight for a kernel-level interference between lanes to show (profiles/r03_stft_concurrency.txt)
    batches = [(0.3 * torch.randn(3, 1, L, generator=gen)).to(dev) for _ in range(NB)]
    lens = [torch.tensor([L, L - 700 * (i % 5 + 1), L // 2], device=dev) for i in range(NB)]
    torch.manual_seed(99)
    ref = []
    for a, l in zip(batches, lens):
        ids, il = codec.encode(a, l)
        wav, _ = codec.decode(ids, il, return_audios=True)
        ref.append((ids.clone(), wav.clone()))
    torch.cuda.synchronize()
    lanes = CodecLanes(codec, 2)
    assert len(lanes) == 2 and lanes.codecs[1] is not codec and lanes.codecs[1].encoder._handle is None      # a copy builds its own handle
    torch.manual_seed(99)
    results = [lanes.roundtrip(a, l) for a, l in zip(batches, lens)]
    assert lanes.codecs[1].encoder._handle is not None and lanes.codecs[1].encoder._handle != codec.encoder._handle
    for (ids_ref, wav_ref), r in zip(ref, results):
        ids, il, wav = r.wait()
        assert torch.equal(ids, ids_ref)
        assert torch.equal(wav, wav_ref)
    lanes.synchronize()
    # weights changed on lane 0 reach the other lanes through refresh()
    with torch.no_grad():
        codec.quality_projection.bias.add_(0.25)
    lanes.refresh()
    assert torch.equal(lanes.codecs[1].quality_projection.bias, codec.quality_projection.bias)


def test_stft_next_to_convolutions_on_another_stream(dev):
    """Regression for the interference of profiles/r03_stft_concurrency.txt: with dmel_stft_set_exclusive_cu(1) (what CodecLanes sets) the STFT
    kernel returns the same bits whether or not conv_bf16_kernel runs on another stream at the same time.  (Without the switch 3-11 % of
    the launches come back with one wrong frame; that is not asserted -- it depends on co-residency -- only the cure is.)"""
    import dmel_codec_amd.torch_ops  # noqa: F401
    from dmel_codec_amd import _lib
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    L = _lib.lib()
    mel_t = LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=80).to(dev)
    gen = torch.Generator().manual_seed(3)
    clips = [(0.3 * torch.randn(3, 24000, generator=gen)).to(dev) for _ in range(4)]
    big = torch.randn(8, 256, 736, generator=gen).to(dev)
    w, b = (torch.randn(256, 256, 3, generator=gen) * 0.05).to(dev), torch.zeros(256, device=dev)
    ref = [mel_t(c).clone() for c in clips]
    y_ref = torch.ops.dmel_hip.conv1d_dilated(big, w, b, 1).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    _lib.check(L.dmel_stft_set_exclusive_cu(1), "stft_set_exclusive_cu")
    try:
        wrong = wrong_y = 0
        for _ in range(40):
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sb):
                ys = [torch.ops.dmel_hip.conv1d_dilated(big, w, b, 1) for _ in range(12)]
            with torch.cuda.stream(sa):
                mels = [mel_t(clips[j % 4]) for j in range(24)]
            torch.cuda.synchronize()
            wrong += sum(0 if torch.equal(m, ref[j % 4]) else 1 for j, m in enumerate(mels))
            wrong_y += sum(0 if torch.equal(y, y_ref) else 1 for y in ys)
        assert wrong == 0 and wrong_y == 0, (wrong, wrong_y)
    finally:
        _lib.check(L.dmel_stft_set_exclusive_cu(0), "stft_set_exclusive_cu")
    torch.cuda.synchronize()
    assert torch.equal(mel_t(clips[0]), ref[0])      # and the regular allocation again


def test_streaming_decode_by_graph_replay_is_bit_identical(dev):
    """VQGAN.decode_stream(use_graph=True): once the stream is in its steady state a push of `chunk_tokens` tokens is one HIP-graph replay
    (quantiser window, every WaveNet block's new columns, vocoder window: ~250 launches -> 1).  The concatenated output must still be
    BIT-identical to decode() of the whole sequence (lm_lit_modules.py:467-471 decodes once, at the end), most pushes must actually have
    been replays, and an odd-sized last chunk + the flush fall back to the launch-by-launch path on the same state."""
    codec = make_codec(710, n_mels=80, dmel_groups=8, encoder_layers=2).to(dev)
    g = torch.Generator().manual_seed(91)
    B, T4, chunk = 2, 333, 24
    ids = torch.randint(0, 175, (B, 8, T4), generator=g, dtype=torch.int32).to(dev)
    noise = torch.randn(B, codec.decoder.input_channels, T4 * 4, generator=g).to(dev)
    flen = torch.full((B,), T4, device=dev)
    wav_ref, mel_ref = codec.decode(ids, flen, return_audios=True, noise=noise)
    for use_graph in (False, True):
        dec = codec.streaming_decoder(B, None, True, graph_chunk_tokens=chunk if use_graph else None)
        wavs, mels = [], []
        for a0 in range(0, T4, chunk):
            n = min(chunk, T4 - a0)
            a, m = dec.push(ids[:, :, a0:a0 + n], noise=noise[:, :, a0 * 4:(a0 + n) * 4])
            wavs.append(a); mels.append(m)
        a, m = dec.finish()
        wavs.append(a); mels.append(m)
        wav, mel = torch.cat(wavs, dim=-1), torch.cat(mels, dim=-1)
        assert wav.shape == wav_ref.shape and torch.equal(mel, mel_ref) and torch.equal(wav, wav_ref), use_graph
        if use_graph:
            report(f"[stream] graph replay: {dec.graph_replays} of {T4 // chunk} full chunks were graph replays")
            assert dec.graph_replays >= T4 // chunk - 8


def test_ragged_batch_with_empty_item(dev):
    """A batch whose second item is shorter than one hop (0 valid frames -> 0 valid tokens): ids and lengths must still
    match the oracle for every item, and decode must emit the all-masked result for the empty one."""
    from dmel_codec_amd.configs import oracle_cfg
    codec = make_codec(900, n_mels=80, dmel_groups=8, encoder_layers=3, decoder_layers=2, vocoder=None)
    cfg = oracle_cfg(codec)
    sd, _ = split_sd(codec)
    gen = torch.Generator().manual_seed(77)
    L = 6000
    audio = torch.randn(3, 1, L, generator=gen) * 0.2
    lens = torch.tensor([L, 100, 3333])
    ids_ref, lens_ref, pre_ref = ref_cpu.vqgan_encode(sd, cfg, audio, lens, return_prequant=True)
    codec = codec.to(dev)
    ids, ilens = codec.encode(audio.to(dev), lens.to(dev))
    assert ilens.tolist() == lens_ref.tolist() == [5, 0, 3]
    n_diff, n_bad, n_tie = near_tie_report(ids, ids_ref, pre_ref)
    assert n_bad == 0
    noise = torch.randn(3, 560, ids.shape[2] * 4, generator=gen)
    mel_ref = ref_cpu.vqgan_decode(sd, cfg, ids_ref, lens_ref, noise)
    mel = codec.decode(ids_ref.to(dev), lens_ref.to(dev), noise=noise.to(dev))
    assert rel_err(mel, mel_ref) < TOL
    assert torch.all(mel[1] == 0)


def test_errors_on_bad_shapes(dev):
    codec = make_codec(910, n_mels=80, dmel_groups=8, encoder_layers=1, decoder_layers=1, vocoder=None).to(dev)
    with pytest.raises((RuntimeError, ValueError)):
        codec.encode(torch.zeros(2, 1, 200, device=dev), torch.tensor([200, 200], device=dev))      # shorter than the reflect pad
    with pytest.raises(ValueError):
        codec.decode(torch.zeros(1, 7, 4, dtype=torch.int32, device=dev), torch.tensor([4], device=dev))   # 7 groups != 8
    with pytest.raises(ValueError):
        codec.decode(torch.zeros(1, 8, 4, dtype=torch.int32, device=dev), torch.tensor([4], device=dev),
                     noise=torch.zeros(1, 560, 15, device=dev))
    with pytest.raises(ValueError, match="Vocoder"):
        codec.decode(torch.zeros(1, 8, 4, dtype=torch.int32, device=dev), torch.tensor([4], device=dev), return_audios=True)


# ------------------------------------------------------------------------------------ opt-in bf16 operand mode
# Not the parity path: these tests pin what the mode IS (operands rounded to bf16 with round-to-nearest-even, exact
# products, fp32 accumulation) and bound how far it moves the outputs, so it cannot drift silently.
@pytest.mark.parametrize("Cout,Cin,k,dil,T,B", [
    (32, 32, 3, 1, 100, 2), (32, 32, 11, 5, 300, 2), (64, 64, 7, 3, 257, 1), (128, 128, 3, 5, 130, 2),
    (256, 256, 11, 1, 96, 1), (140, 70, 3, 8, 93, 3), (512, 100, 7, 1, 92, 2), (70, 280, 1, 1, 23, 5),
    (100, 700, 1, 1, 92, 1), (33, 17, 5, 2, 65, 1), (128, 48, 7, 9, 2500, 1),
])
def test_conv_bf16_operand_mode(dev, Cout, Cin, k, dil, T, B):
    from dmel_codec_amd import _lib
    torch.manual_seed(Cout * 1000 + Cin + k)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout) * 0.1
    x = torch.randn(B, Cin, T)
    ref = F.conv1d(x.bfloat16().double(), w.bfloat16().double(), b.double(), dilation=dil, padding=dil * (k - 1) // 2)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cout, Cin, k, dil))
    try:
        assert L.dmel_conv_set_precision(h, 7) < 0
        _lib.check(L.dmel_conv_set_precision(h, 1))
        xd = x.to(dev)
        y = torch.empty(B, Cout, T, device=dev)
        _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert rel_err(y, ref) < 1e-5          # only fp32 accumulation order separates the two
        _lib.check(L.dmel_conv_set_precision(h, 0))                     # and back: the fp32 path is untouched
        _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
        assert rel_err(y, F.conv1d(x, w, b, dilation=dil, padding=dil * (k - 1) // 2)) < 2e-5
    finally:
        L.dmel_conv_destroy(h)


def test_decode_bf16_mode_stays_close_to_fp32(dev):
    codec = make_codec(320, n_mels=80, dmel_groups=8, decoder_layers=6, encoder_layers=2)
    with torch.no_grad():
        codec.vocoder.conv_post.weight_g.fill_(0.02)     # keep the random vocoder out of tanh saturation: errors stay readable
    codec = codec.to(dev)
    B, T4 = 3, 40
    gen = torch.Generator().manual_seed(99)
    ids = torch.randint(0, 175, (B, 8, T4), generator=gen, dtype=torch.int32).to(dev)
    flen = torch.tensor([T4, 17, 33], device=dev)
    noise = torch.randn(B, 560, T4 * 4, generator=gen).to(dev)
    audio32, mel32 = codec.decode(ids, flen, return_audios=True, noise=noise)
    codec.set_decode_precision("bf16")
    audio16, mel16 = codec.decode(ids, flen, return_audios=True, noise=noise)
    assert not torch.equal(mel16, mel32)                      # the mode really ran
    assert rel_err(mel16, mel32) < 1e-2, rel_err(mel16, mel32)
    assert rel_err(audio16, audio32) < 0.15, rel_err(audio16, audio32)   # random deep net: amplifies; a trained vocoder does not
    assert torch.all(mel16[1, :, 17 * 4:] == 0)               # masks are applied exactly as in fp32
    # encode() ignores the mode: ids stay bit-stable
    a = torch.randn(2, 1, 12000, generator=gen).to(dev) * 0.1
    l = torch.tensor([12000, 9000], device=dev)
    ids16, _ = codec.encode(a, l)
    codec.set_decode_precision("fp32")
    ids32, _ = codec.encode(a, l)
    assert torch.equal(ids16, ids32)
    audio_back, mel_back = codec.decode(ids, flen, return_audios=True, noise=noise)
    assert torch.equal(mel_back, mel32) and torch.equal(audio_back, audio32)
    with pytest.raises(ValueError):
        codec.decoder.set_precision("fp16")
    with pytest.raises(NotImplementedError):
        codec.quantizer.set_precision("bf16")


@pytest.mark.parametrize("Cout,Cin,k,dil,T,B", [(256, 256, 7, 3, 736, 2), (128, 128, 11, 5, 3000, 1), (1120, 560, 3, 2, 92, 2),
                                                (64, 64, 7, 1, 1500, 2), (32, 32, 11, 5, 4000, 1), (140, 70, 3, 4, 93, 4)])
def test_split_fp32_conv_is_fp32_grade(dev, Cout, Cin, k, dil, T, B):
    """The default fp32 path forms products from an exact 3-way bf16 split on the bf16 matrix cores.  Against an fp64
    evaluation its error must be no worse than that of the native fp32 MFMA kernel (DMEL_PRECISION_FP32_MFMA) by more than
    rounding noise, and both must sit at fp32 accumulation level."""
    from dmel_codec_amd import _lib
    torch.manual_seed(Cout + Cin + k + T)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout) * 0.1
    x = torch.randn(B, Cin, T) * torch.logspace(-3, 1, Cin)[None, :, None]      # channels spanning four decades
    ref = F.conv1d(x.double(), w.double(), b.double(), dilation=dil, padding=dil * (k - 1) // 2)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cout, Cin, k, dil))
    try:
        xd = x.to(dev)
        errs = {}
        for name, mode in (("split", 0), ("native", 2), ("f16x2", 3), ("bf16x3", 4)):
            _lib.check(L.dmel_conv_set_precision(h, mode))
            y = torch.empty(B, Cout, T, device=dev)
            _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
            torch.cuda.synchronize()
            errs[name] = rel_err(y, ref)
            if name == "bf16x3":
                assert errs[name] == errs["split"]
        assert errs["native"] < 2e-6 and errs["split"] < 2e-6 and errs["f16x2"] < 2e-6, errs
        assert errs["split"] < 1.5 * errs["native"] + 1e-7, errs
        assert errs["f16x2"] < 1.5 * errs["native"] + 1e-7, errs     # three fp16 products: the same fp32-grade result
        from conftest import report
        report(f"conv {Cout}x{Cin} k{k} d{dil} T{T}: error vs fp64  fp32-MFMA {errs['native']:.2e}  bf16x3 (6 products) {errs['split']:.2e}  "
               f"f16x2 (3 products) {errs['f16x2']:.2e}")
    finally:
        L.dmel_conv_destroy(h)


@pytest.mark.parametrize("scale", [1e-7, 1e-5, 1e-3, 1.0, 3e3, 5e5])
def test_f16_split_conv_over_the_magnitude_range(dev, scale):
    """DMEL_PRECISION_FP32_F16X2 (include/dmel_hip.h): the second fp16 piece is kept scaled by 2^11 and inputs are staged x 2^-6, so the
    relative error stays at fp32 level from |x| ~ 1e-3 up to 4e6 and becomes an ABSOLUTE 2^-30 below |x| = 2^-8 (fp16 subnormals of the
    first piece; the matrix core must not flush them)."""
    from dmel_codec_amd import _lib
    Cout, Cin, k, dil, T, B = 128, 96, 7, 3, 500, 2
    torch.manual_seed(int(abs(math.log10(scale)) * 10) + 3)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    x = torch.randn(B, Cin, T) * scale
    ref = F.conv1d(x.double(), w.double(), None, dilation=dil, padding=dil * (k - 1) // 2)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), None, Cout, Cin, k, dil))
    try:
        xd = x.to(dev)
        _lib.check(L.dmel_conv_set_precision(h, 3))
        y = torch.empty(B, Cout, T, device=dev)
        _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
        torch.cuda.synchronize()
        err_abs = (y.double().cpu() - ref).abs().max().item()
        # fp32-level relative error, or the absolute floor: K = 672 terms of <= 2^-30 |w| each, |w| ~ 0.04 (measured 2.5e-9 at |x| ~ 1e-7)
        assert err_abs < max(2e-6 * ref.abs().max().item(), 2.0 ** -30 * 4.0), (scale, err_abs, ref.abs().max().item())
        from conftest import report
        report(f"f16x2 conv, |x| ~ {scale:g}: max abs error {err_abs:.3e} (max |y| {ref.abs().max().item():.3e})")
        if scale == 5e5:      # beyond 2^6 * 65504 = 4.19e6 the first piece overflows: the result is non-finite, never a silently wrong number
            xd[0, 3, 100] = 1e7
            _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert not torch.isfinite(y[0, :, 100 - 9:100 + 10]).all()
            _lib.check(L.dmel_conv_set_precision(h, 0))          # the six-product split has the range of fp32
            _lib.check(L.dmel_conv_forward(h, xd.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert torch.isfinite(y).all()
    finally:
        L.dmel_conv_destroy(h)


@pytest.mark.parametrize("dy_scale", [1e-12, 1e-7, 1e-3, 1.0, 1e4])
@pytest.mark.parametrize("Cout,Cin,k,dil,T,B", [(64, 48, 7, 3, 1000, 2), (128, 96, 3, 1, 515, 3)])
def test_conv_weight_gradient_fp16_split_at_any_gradient_magnitude(dev, Cout, Cin, k, dil, T, B, dy_scale):
    """Rows of >= 256 samples take the fp16-split weight-gradient kernel (conv_bwd.hip): dy is staged x 2^(13 - exponent of max |dy|), found by a
    reduction on the same stream, so the result is fp32-grade whatever the size of the gradients -- including entries eight decades below
    the largest one."""
    from dmel_codec_amd import _lib
    torch.manual_seed(Cout + Cin + T)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    x = torch.randn(B, Cin, T)
    dy = torch.randn(B, Cout, T) * dy_scale * torch.logspace(-8, 0, Cout)[None, :, None]      # rows spanning eight decades
    w64, x64 = w.double().requires_grad_(), x.double().requires_grad_()
    F.conv1d(x64, w64, None, dilation=dil, padding=dil * (k - 1) // 2).backward(dy.double())
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), None, Cout, Cin, k, dil))
    try:
        xd, dyd = x.to(dev), dy.to(dev)
        dw = torch.full((Cout, Cin, k), float("nan"), device=dev)
        db = torch.full((Cout,), float("nan"), device=dev)
        _lib.check(L.dmel_conv_backward_weight(h, xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), B, T, _lib.stream_ptr()))
        torch.cuda.synchronize()
        ref = w64.grad
        assert rel_err(dw, ref) < 2e-6
        # row by row: the small rows are exact relative to THEIR OWN size down to 2^-28 of the largest gradient (here 1e-8 of it: the smallest row
        # sits at the edge and is allowed 1e-4; from 1e-6 of the maximum upwards fp32-grade)
        for co in range(0, Cout, 7):
            bar = 2e-6 if co >= Cout // 4 else 1e-4
            assert rel_err(dw[co], ref[co]) < bar, (co, rel_err(dw[co], ref[co]))
    finally:
        L.dmel_conv_destroy(h)


# ------------------------------------------------------------------------------------ conv backward (training path, first piece)
@pytest.mark.parametrize("Cout,Cin,k,dil,T,B", [
    (32, 32, 3, 1, 100, 2), (64, 48, 7, 3, 257, 2), (140, 70, 3, 4, 93, 5), (70, 140, 1, 1, 93, 5), (128, 128, 11, 5, 700, 1),
    (33, 17, 5, 2, 65, 3), (1120, 560, 3, 2, 92, 2), (100, 560, 1, 1, 92, 3),
])
def test_conv_backward_matches_autograd(dev, Cout, Cin, k, dil, T, B):
    """dmel_conv_backward_data / _weight against torch autograd of F.conv1d evaluated in float64 (what the reference's
    loss.backward() computes for every ConvNorm / Conv1d of the WaveNets, codec_lit_modules.py:236,315)."""
    from dmel_codec_amd import _lib
    torch.manual_seed(Cout + 7 * Cin + k + T)
    w = torch.randn(Cout, Cin, k) / math.sqrt(Cin * k)
    b = torch.randn(Cout) * 0.1
    x = torch.randn(B, Cin, T)
    dy = torch.randn(B, Cout, T)
    w64, b64, x64 = w.double().requires_grad_(), b.double().requires_grad_(), x.double().requires_grad_()
    F.conv1d(x64, w64, b64, dilation=dil, padding=dil * (k - 1) // 2).backward(dy.double())
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), b.data_ptr(), Cout, Cin, k, dil))
    try:
        xd, dyd = x.to(dev), dy.to(dev)
        dx = torch.empty(B, Cin, T, device=dev)
        dw = torch.full((Cout, Cin, k), float("nan"), device=dev)        # must be overwritten, not accumulated into
        db = torch.full((Cout,), float("nan"), device=dev)
        st = _lib.stream_ptr()
        _lib.check(L.dmel_conv_backward_data(h, dyd.data_ptr(), dx.data_ptr(), B, T, st))
        _lib.check(L.dmel_conv_backward_weight(h, xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), B, T, st))
        torch.cuda.synchronize()
        assert rel_err(dx, x64.grad) < 2e-6
        assert rel_err(dw, w64.grad) < 2e-6
        assert rel_err(db, b64.grad) < 2e-6
        _lib.check(L.dmel_conv_backward_weight(h, xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), None, B, T, st))   # db optional
        torch.cuda.synchronize()
        assert rel_err(dw, w64.grad) < 2e-6
        assert L.dmel_conv_backward_data(h, None, dx.data_ptr(), B, T, st) < 0
    finally:
        L.dmel_conv_destroy(h)


@pytest.mark.parametrize("cfg", [
    dict(input_channels=10, residual_channels=70, residual_layers=5, dilation_cycle=4, N=6, T=93),                     # encoder-like
    dict(input_channels=48, output_channels=20, residual_channels=48, residual_layers=4, dilation_cycle=4,
         condition_channels=48, N=3, T=61),                                                                               # decoder-like
    dict(residual_channels=64, residual_layers=3, dilation_cycle=2, N=2, T=130),                                         # no projections
    dict(input_channels=560, output_channels=80, residual_channels=560, residual_layers=2, dilation_cycle=4,
         condition_channels=560, N=2, T=92),                                                                              # real decoder widths
])
def test_wavenet_training_forward_backward(dev, cfg):
    """Native forward_train / backward of the WaveNet against autograd through the oracle in float64: output, input and
    condition gradients and the gradient of every parameter (what manual_backward produces in training_step,
    codec_lit_modules.py:236,315).  Also: forward_train == inference forward, and the dead diffusion weights get no grad."""
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    cfg = dict(cfg)
    N, T = cfg.pop("N"), cfg.pop("T")
    torch.manual_seed(T + N)
    m = WaveNet(**cfg)
    randomise(m, 40 + N)
    sd = cpu_sd(m)
    sd64 = {k: v.double().requires_grad_() for k, v in sd.items()}
    cin = m.input_channels
    x = torch.randn(N, cin, T)
    cond = torch.randn(N, cfg["condition_channels"], T) if cfg.get("condition_channels") else None
    gy = torch.randn(N, m.output_channels, T)
    x64 = x.double().requires_grad_()
    c64 = cond.double().requires_grad_() if cond is not None else None
    y64 = ref_cpu.wavenet_forward(sd64, "", x64, len(m.residual_layers), m.dilation_cycle or 0, c64)
    (y64 * gy.double()).sum().backward()

    m = m.to(dev)
    xd = x.to(dev).requires_grad_()
    cd = cond.to(dev).requires_grad_() if cond is not None else None
    y = m(xd, condition=cd)
    assert y.requires_grad
    with torch.no_grad():
        y_inf = m(xd.detach(), condition=cd.detach() if cd is not None else None)
    assert rel_err(y, y64) < 2e-5 and rel_err(y_inf, y64) < 2e-5
    (y * gy.to(dev)).sum().backward()
    assert rel_err(xd.grad, x64.grad) < 2e-5
    if cond is not None:
        assert rel_err(cd.grad, c64.grad) < 2e-5
    checked = 0
    for k, p in m.named_parameters():
        if "diffusion_projection" in k:
            assert p.grad is None
            continue
        assert p.grad is not None, k
        assert rel_err(p.grad, sd64[k].grad) < 2e-5, (k, rel_err(p.grad, sd64[k].grad))
        checked += 1
    assert checked == len([k for k in sd if "diffusion_projection" not in k])
    # a second step after an optimiser-style in-place update rebuilds the handle and still matches
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01 * torch.randn_like(p))
    m.zero_grad()
    y2 = m(xd.detach(), condition=cd.detach() if cd is not None else None)
    sd2 = {k: v.double() for k, v in cpu_sd(m).items()}
    assert rel_err(y2, ref_cpu.wavenet_forward(sd2, "", x.double(), len(m.residual_layers), m.dilation_cycle or 0,
                                               cond.double() if cond is not None else None)) < 2e-5
    y2.sum().backward()
    assert m.skip_projection.conv.weight.grad is not None
    # ... and that update went through the device-side re-pack (same handle), whose images are bit-identical to a rebuild
    h_before = m._handle
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)
        y3 = m(xd.detach(), condition=cd.detach() if cd is not None else None)
        assert m._handle == h_before
        m._free_native()
        y4 = m(xd.detach(), condition=cd.detach() if cd is not None else None)
        assert m._handle is not None and torch.equal(y3, y4)
    g3 = torch.autograd.grad(m(xd.detach(), condition=cd.detach() if cd is not None else None).sum(), m.skip_projection.conv.weight)[0]
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.0)                    # version bump, same values: refresh path again
    g4 = torch.autograd.grad(m(xd.detach(), condition=cd.detach() if cd is not None else None).sum(), m.skip_projection.conv.weight)[0]
    assert rel_err(g4, g3) < 1e-6          # (wgrad uses atomics: equal to rounding, not bitwise)


def test_wavenet_three_adamw_steps_match_cpu_training(dev):
    """A small end-to-end training loop on the native path -- masked mel-L1 loss as in training_step
    (codec_lit_modules.py:247-263), torch AdamW, three steps, weights re-packed on the device between steps -- against the same
    loop run on the CPU through the oracle with autograd in float64: losses and final parameters must agree."""
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    torch.manual_seed(5)
    kw = dict(input_channels=32, output_channels=16, residual_channels=32, residual_layers=3, dilation_cycle=2, condition_channels=32)
    m = WaveNet(**kw)
    randomise(m, 77)
    N, T = 4, 50
    lens = torch.tensor([50, 31, 44, 9])
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :]
    noise, cond, target = torch.randn(N, 32, T), torch.randn(N, 32, T), torch.randn(N, 16, T)

    def loss_fn(y, tgt, msk):              # avg_with_mask(|gen - gt|, mask)   utils/utils.py:58-67
        d = (y * msk - tgt * msk).abs()
        return (d * msk).sum() / (msk.sum() * d.shape[1])

    # CPU reference loop (float64 autograd through the oracle)
    sd = {k: v.double().clone().requires_grad_() for k, v in cpu_sd(m).items() if "diffusion_projection" not in k}
    opt_ref = torch.optim.AdamW(list(sd.values()), lr=1e-2, betas=(0.8, 0.99), eps=1e-5)      # dMel_example.yaml optimizer block
    ref_losses = []
    for _ in range(3):
        opt_ref.zero_grad()
        y = ref_cpu.wavenet_forward(sd, "", noise.double() * mask.double(), 3, 2, cond.double() * mask.double())
        l = loss_fn(y, target.double(), mask.double())
        l.backward()
        opt_ref.step()
        ref_losses.append(l.item())

    m = m.to(dev)
    named = [(k, p) for k, p in m.named_parameters() if "diffusion_projection" not in k]
    opt = torch.optim.AdamW([p for _, p in named], lr=1e-2, betas=(0.8, 0.99), eps=1e-5)
    nd, cdv, td, md = noise.to(dev), cond.to(dev), target.to(dev), mask.to(dev)
    losses = []
    handle = None
    for step in range(3):
        opt.zero_grad()
        y = m(nd * md, condition=cdv * md)
        l = loss_fn(y, td, md)
        l.backward()
        opt.step()
        losses.append(l.item())
        if step == 0:
            handle = m._handle
    assert m._handle == handle                       # the optimiser steps went through the device-side re-pack
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-5 * abs(b), (losses, ref_losses)
    assert ref_losses[2] < ref_losses[0]
    for k, p in named:
        assert rel_err(p.detach(), sd[k].detach()) < 5e-5, k


@pytest.mark.parametrize("B,Cc,T", [(2, 5, 1023), (1, 3, 1024), (2, 2, 1025), (1, 4, 5000), (3, 16, 333), (2, 3, 1), (1, 2, 2), (1, 2, 7)])
@pytest.mark.parametrize("kind", ["snakebeta_log", "snakebeta_lin", "snake_log"])
def test_activation1d_backward_matches_autograd(dev, B, Cc, T, kind):
    """dmel_aa_snake_backward_f32 (the reference's fused kernel has no backward; its torch path trains through autograd)
    against autograd through the oracle's six-op Activation1d in float64: d input and the gradients of the stored alpha / beta."""
    from dmel_codec_amd import _lib
    torch.manual_seed(T * 7 + Cc)
    logscale = kind.endswith("_log")
    snake = kind.startswith("snake_")
    x = torch.randn(B, Cc, T) * 2
    dy = torch.randn(B, Cc, T)
    alpha = torch.randn(Cc) * 0.5 if logscale else torch.rand(Cc) + 0.5
    beta = None if snake else (torch.randn(Cc) * 0.5 if logscale else torch.rand(Cc) + 0.5)
    x64, a64 = x.double().requires_grad_(), alpha.double().requires_grad_()
    b64 = beta.double().requires_grad_() if beta is not None else None
    y64 = ref_cpu.activation1d(x64, a64, b64 if b64 is not None else a64, ref_cpu.aa_filter12().double(), None, logscale=logscale, snake=snake)
    (y64 * dy.double()).sum().backward()
    taps = ref_cpu.aa_filter12().view(-1).contiguous()
    xd, dyd, ad = x.to(dev), dy.to(dev), alpha.to(dev)
    bd = beta.to(dev) if beta is not None else None
    dx = torch.full((B, Cc, T), float("nan"), device=dev)
    da = torch.full((Cc,), float("nan"), device=dev)
    db = torch.full((Cc,), float("nan"), device=dev) if beta is not None else None
    L = _lib.lib()
    _lib.check(L.dmel_aa_snake_backward_f32(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), ad.data_ptr(), _lib.ptr(bd), da.data_ptr(),
                                            _lib.ptr(db), taps.data_ptr(), taps.data_ptr(), int(logscale), B, Cc, T, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert rel_err(dx, x64.grad) < 2e-5
    assert rel_err(da, a64.grad) < 1e-4, (da, a64.grad)
    if beta is not None:
        assert rel_err(db, b64.grad) < 1e-4
    # beta / dbeta must be given together
    assert L.dmel_aa_snake_backward_f32(xd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), ad.data_ptr(), None, da.data_ptr(),
                                        da.data_ptr(), taps.data_ptr(), taps.data_ptr(), int(logscale), B, Cc, T, _lib.stream_ptr()) < 0


def test_activation1d_module_is_differentiable(dev):
    from dmel_codec_amd.models.modules.bigvgan.alias_free_activation.act import Activation1d
    from dmel_codec_amd.models.modules.bigvgan import activations
    torch.manual_seed(3)
    m = Activation1d(activation=activations.SnakeBeta(6, alpha_logscale=True)).to(dev)
    with torch.no_grad():
        m.act.alpha.normal_(0, 0.3)
        m.act.beta.normal_(0, 0.3)
    x = torch.randn(2, 6, 300, device=dev, requires_grad=True)
    g = torch.randn(2, 6, 300, device=dev)
    y = m(x)
    (y * g).sum().backward()
    x64 = x.detach().cpu().double().requires_grad_()
    a64, b64 = m.act.alpha.detach().cpu().double().requires_grad_(), m.act.beta.detach().cpu().double().requires_grad_()
    y64 = ref_cpu.activation1d(x64, a64, b64, ref_cpu.aa_filter12().double(), None, logscale=True)
    (y64 * g.cpu().double()).sum().backward()
    assert rel_err(y, y64) < 1e-5 and rel_err(x.grad, x64.grad) < 2e-5
    assert rel_err(m.act.alpha.grad, a64.grad) < 1e-4 and rel_err(m.act.beta.grad, b64.grad) < 1e-4
    with torch.no_grad():
        assert torch.equal(m(x.detach()), y.detach())


@pytest.mark.parametrize("dim,N,T", [(70, 6, 46), (70, 3, 23), (16, 2, 100), (48, 4, 33)])
def test_convnext_block_forward_backward(dev, dim, N, T):
    """Standalone ConvNeXtBlock: inference forward, forward_train and the hand-written backward (LayerNorm-over-channels, depthwise
    conv, exact-erf GELU, layer scale) against autograd through the oracle in float64 -- input gradient and all nine parameters."""
    from dmel_codec_amd.models.modules.firefly import ConvNeXtBlock
    torch.manual_seed(dim + T)
    m = ConvNeXtBlock(dim)
    randomise(m, 11 + dim)
    with torch.no_grad():
        m.gamma.normal_(0, 0.5)          # the reference initialises gamma to 1e-6, which would hide every inner gradient
        m.norm.weight.normal_(1, 0.3)
        m.norm.bias.normal_(0, 0.3)
    sd64 = {k: v.double().requires_grad_() for k, v in cpu_sd(m).items()}
    x = torch.randn(N, dim, T)
    gy = torch.randn(N, dim, T)
    x64 = x.double().requires_grad_()
    y64 = ref_cpu.convnext_block(sd64, "", x64)
    (y64 * gy.double()).sum().backward()
    m = m.to(dev)
    xd = x.to(dev).requires_grad_()
    with torch.no_grad():
        y_inf = m(xd.detach())
    y = m(xd)
    assert rel_err(y_inf, y64) < 2e-5 and rel_err(y, y64) < 2e-5
    (y * gy.to(dev)).sum().backward()
    assert rel_err(xd.grad, x64.grad) < 2e-5
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, sd64[k].grad) < 5e-5, (k, rel_err(p.grad, sd64[k].grad))


@pytest.mark.parametrize("levels,prebound,G,T,B", [([7, 5, 5], True, 3, 93, 2), ([7, 5, 5], False, 2, 64, 3), ([8, 6], True, 2, 47, 2)])
def test_quantizer_training_forward_backward(dev, levels, prebound, G, T, B):
    """DownsampleFiniteScalarQuantize.forward (straight-through FSQ, pad back to T) and its hand-written backward against autograd
    through the oracle in float64: output, ids, input gradient and the gradient of all 13 * 4 + 4 * G parameter tensors."""
    from dmel_codec_amd.models.modules.dowmsample_fsq import DownsampleFiniteScalarQuantize
    torch.manual_seed(T + G)
    Cg = 70
    q = DownsampleFiniteScalarQuantize(input_dim=Cg * G, n_codebooks=1, n_groups=G, levels=levels, downsample_factor=[2, 2],
                                       is_dmel=True, fsq_prebound=prebound)
    randomise(q, 91 + G, scale=1.5)
    with torch.no_grad():
        for m in q.modules():
            if hasattr(m, "gamma"):
                m.gamma.normal_(0, 0.5)
    q = q.to(dev)
    # A latent within rounding noise of a quantisation boundary may legitimately round the other way, which changes the whole
    # straight-through gradient: the input is re-drawn (deterministically) until the float64 oracle keeps a 1e-4 margin to every
    # boundary -- then ids must be EQUAL and the gradient check always runs (no skip).
    gen = torch.Generator().manual_seed(1000 * T + G)
    for attempt in range(20):
        z = torch.randn(B * G, Cg, T, generator=gen)
        sd64 = {k: v.double().requires_grad_() for k, v in cpu_sd(q).items()}
        z64 = z.double().requires_grad_()
        zq64, ids64, lat64 = ref_cpu.quantizer_forward(sd64, "", z64, G, levels, (2, 2), prebound)
        _, pre64 = ref_cpu.quantizer_encode({k: v.detach() for k, v in sd64.items()}, "", z.double(), G, levels, (2, 2), prebound,
                                            return_prequant=True)
        if float((pre64 - torch.floor(pre64) - 0.5).abs().min()) > 1e-4:
            break
    else:
        raise AssertionError("no input with a 1e-4 rounding margin in 20 draws")
    gz = torch.randn(B, Cg * G, T, generator=gen)
    (zq64 * gz.double()).sum().backward()
    zd = z.to(dev).requires_grad_()
    res = q(zd)
    assert res.z.shape == zq64.shape and res.codes.shape == ids64.shape and res.latents.shape == lat64.shape
    assert rel_err(res.latents, lat64) < 2e-5
    assert torch.equal(res.codes.cpu(), ids64.cpu().to(res.codes.dtype)), int((res.codes.cpu() != ids64.cpu()).sum())
    assert rel_err(res.z, zq64) < 5e-5
    (res.z * gz.to(dev)).sum().backward()
    assert rel_err(zd.grad, z64.grad) < 1e-4
    for k, p in q.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, sd64[k].grad) < 2e-4, (k, rel_err(p.grad, sd64[k].grad))
    # optimiser-style in-place update: the handle is re-packed on the device, bit-identically to a rebuild through the host
    h_before = q._handle
    with torch.no_grad():
        for p in q.parameters():
            p.mul_(1.003)
        r1 = q(zd.detach())
        ids1 = q.encode(zd.detach())
        dec1 = q.decode(ids1)
        assert q._handle == h_before
        q._free_native()
        r2 = q(zd.detach())
        assert torch.equal(r1.z, r2.z) and torch.equal(r1.codes, r2.codes) and torch.equal(r1.latents, r2.latents)
        assert torch.equal(q.encode(zd.detach()), ids1) and torch.equal(q.decode(ids1), dec1)
    g1 = torch.autograd.grad(q(zd.detach()).z.sum(), q.downsample[0][0].weight)[0]
    with torch.no_grad():
        for p in q.parameters():
            p.mul_(1.0)
    g2 = torch.autograd.grad(q(zd.detach()).z.sum(), q.downsample[0][0].weight)[0]
    assert rel_err(g2, g1) < 1e-6


def test_generator_half_of_training_step(dev):
    """VQGAN.generator_forward + mel_loss (codec_lit_modules.py:164-211, 246-263) on the native training paths -- STFT front end,
    grouped encoder WaveNet, straight-through quantiser, quality projection, conditioned decoder WaveNet, masks, band-weighted L1 --
    against the same chain through the oracle with autograd in float64: loss, gen_mel and the gradient of EVERY trainable parameter
    of encoder, quantiser, decoder and quality_projection.  Ragged lengths; the decoder noise is injected."""
    from dmel_codec_amd.configs import oracle_cfg
    codec = make_codec(1234, n_mels=80, dmel_groups=8, encoder_layers=3, decoder_layers=2, vocoder=None)
    with torch.no_grad():
        for m in codec.quantizer.modules():
            if hasattr(m, "gamma"):
                m.gamma.normal_(0, 0.3)
    cfg = oracle_cfg(codec)
    sd, _ = split_sd(codec)
    gen = torch.Generator().manual_seed(11)
    L = 9000
    audio = torch.randn(3, 1, L, generator=gen) * 0.2
    lens = torch.tensor([L, 5000, 7777])
    T = L // 256
    noise = torch.randn(3, 560, T, generator=gen)
    sd64 = {k: (v.double().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
    loss64, gen64, gt64 = ref_cpu.vqgan_generator_loss(sd64, cfg, audio, lens, noise)
    loss64.backward()

    codec = codec.to(dev)
    gen_mel, gt_mels, masks = codec.generator_forward(audio.to(dev), lens.to(dev), noise=noise.to(dev))
    loss = codec.mel_loss(gen_mel, gt_mels, masks)
    assert rel_err(gt_mels, gt64) < 1e-4
    assert rel_err(gen_mel, gen64) < 1e-4
    assert abs(loss.item() - loss64.item()) < 1e-4 * abs(loss64.item())
    loss.backward()
    n_checked = 0
    for k, p in codec.named_parameters():
        if "diffusion_projection" in k or k.startswith("vocoder."):
            continue
        ref = sd64[k].grad
        assert p.grad is not None and ref is not None, k
        scale = ref.abs().max().item()
        if scale < 1e-12:
            assert p.grad.abs().max().item() < 1e-9, k
        else:
            assert rel_err(p.grad, ref) < 2e-3, (k, rel_err(p.grad, ref))
        n_checked += 1
    assert n_checked >= 100


def test_validation_step(dev):
    """VQGAN.validation_step (codec_lit_modules.py:330-396): val_loss, the re-synthesised mel (quality input fixed at 2, condition not
    re-masked) and both vocoder outputs against the same statements through the oracle; ragged lengths, decoder noise injected."""
    from dmel_codec_amd.configs import oracle_cfg
    codec = make_codec(77, n_mels=80, dmel_groups=8, encoder_layers=3, decoder_layers=2)
    cfg = oracle_cfg(codec)
    sd, voc = split_sd(codec)
    gen = torch.Generator().manual_seed(5)
    L = 4096
    audio = torch.randn(2, 1, L, generator=gen) * 0.2
    lens = torch.tensor([[L, 3000]])                     # the collate function's (1, B) shape
    noise = torch.randn(2, 560, L // 256, generator=gen)
    h = dict(codec.vocoder.h)
    loss_ref, gen_ref, gt_ref, recon_ref, aux_ref = ref_cpu.vqgan_validation(sd, cfg, audio, lens, noise, voc, h)
    loss64, gen64, _, recon64, aux64 = ref_cpu.vqgan_validation(to64(sd), cfg, audio, lens, noise.double(), to64(voc), h)
    codec = codec.to(dev)
    batch = {"audios": audio.to(dev), "audio_lengths": lens.to(dev)}
    out = codec.validation_step(batch, 0, noise=noise.to(dev))
    assert rel_err(out["gt_mels"], gt_ref) < TOL
    assert_close_to_truth(out["gen_aux_mels"], gen_ref, gen64, "validation mel")
    assert abs(out["val_loss"].item() - loss64.item()) < 1e-4 * abs(loss64.item())
    assert abs(codec.logged["val_loss"] - loss64.item()) < 1e-4 * abs(loss64.item())
    assert out["recon_audios"].shape == recon_ref.shape and out["gen_aux_audios"].shape == aux_ref.shape
    assert_close_to_truth(out["recon_audios"], recon_ref, recon64, "vocoder(gt_mels)")
    assert_close_to_truth(out["gen_aux_audios"], aux_ref, aux64, "vocoder(gen_aux_mels)")
    codec.vocoder = None
    with pytest.raises(ValueError, match="Vocoder is not loaded"):
        codec.validation_step(batch, 0, noise=noise.to(dev))


@pytest.mark.parametrize("B,H,W", [(2, 80, 37), (3, 100, 93), (1, 80, 12), (1, 80, 60), (1, 16, 200), (2, 3, 5)])
def test_discriminator_forward(dev, B, H, W):
    """Native Discriminator forward (every (3, kw) Conv2d as three implicit-GEMM launches over displaced views of the flattened,
    zero-padded image; stride-2 layers as two polyphase K segments) against the oracle, whose restatement is pinned to the reference class by a CPU fixture."""
    from dmel_codec_amd.models.modules.discriminator import Discriminator
    sd = ref_cpu.seeded_discriminator_sd(4242 + W)
    d = Discriminator()
    d.load_state_dict(sd)
    torch.manual_seed(W)
    x = torch.randn(B, H, W)
    ref = ref_cpu.discriminator_forward(sd, "", x)
    ref64 = ref_cpu.discriminator_forward({k: v.double() for k, v in sd.items()}, "", x.double())
    d = d.to(dev)
    with torch.no_grad():
        y = d(x.to(dev))
    assert y.shape == ref.shape
    assert_close_to_truth(y, ref, ref64, "discriminator logits")


@pytest.mark.parametrize("B,H,W", [(2, 80, 37), (1, 80, 93), (1, 16, 60), (1, 8, 300)])
def test_discriminator_backward(dev, B, H, W):
    """Native discriminator backward (transposed convs per kernel row / output phase, multi-tap strided weight-gradient GEMMs, chain through
    weight norm) against autograd through the oracle in float64 (itself pinned to the reference class's autograd by a CPU fixture):
    d input, and bias / weight-norm g / v gradients of all six layers."""
    from dmel_codec_amd.models.modules.discriminator import Discriminator
    sd = ref_cpu.seeded_discriminator_sd(777 + W)
    d = Discriminator()
    d.load_state_dict(sd)
    torch.manual_seed(W + 1)
    x = torch.randn(B, H, W)
    sd64 = {k: v.double().requires_grad_() for k, v in sd.items()}
    x64 = x.double().requires_grad_()
    y64 = ref_cpu.discriminator_forward(sd64, "", x64)
    dy = torch.randn(y64.shape)
    (y64 * dy.double()).sum().backward()
    d = d.to(dev)
    xd = x.to(dev).requires_grad_()
    y = d(xd)
    assert rel_err(y, y64) < 1e-4
    (y * dy.to(dev)).sum().backward()
    assert rel_err(xd.grad, x64.grad) < 1e-4, rel_err(xd.grad, x64.grad)
    for k, p in d.named_parameters():
        assert p.grad is not None, k
        assert rel_err(p.grad, sd64[k].grad) < 2e-4, (k, rel_err(p.grad, sd64[k].grad))


def test_full_training_step_matches_cpu_reference_loop(dev):
    """VQGAN.training_step (codec_lit_modules.py:159-327) end to end on the native training paths -- discriminator step (LSGAN, masked,
    clip, AdamW, scheduler), generator step (mel L1 + adversarial through the just-updated discriminator, clip, AdamW, scheduler) --
    for two consecutive batches, against the same statement sequence executed on the CPU through the oracle with autograd in float64:
    logged losses and every parameter of generator and discriminator after the steps."""
    from functools import partial
    from dmel_codec_amd.configs import oracle_cfg
    from dmel_codec_amd.utils.schedule import get_cosine_schedule_with_warmup_lr_lambda
    opt = partial(torch.optim.AdamW, lr=2e-3, betas=(0.8, 0.99), eps=1e-5)
    sched = partial(torch.optim.lr_scheduler.LambdaLR,
                    lr_lambda=partial(get_cosine_schedule_with_warmup_lr_lambda, num_warmup_steps=1, num_training_steps=10, final_lr_ratio=0.1))
    codec = make_codec(4321, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=2, vocoder=None, discriminator=True,
                       optimizer=opt, lr_scheduler=sched)
    codec.discriminator.load_state_dict(ref_cpu.seeded_discriminator_sd(31337))
    with torch.no_grad():
        for m in codec.quantizer.modules():
            if hasattr(m, "gamma"):
                m.gamma.normal_(0, 0.3)
    cfg = oracle_cfg(codec)
    full_sd = cpu_sd(codec)
    gsd = {k: v.double().requires_grad_() for k, v in full_sd.items()
           if not k.startswith(("discriminator.", "vocoder.")) and "diffusion_projection" not in k and v.is_floating_point()}
    dsd = {k[len("discriminator."):]: v.double().requires_grad_() for k, v in full_sd.items() if k.startswith("discriminator.")}
    gen = torch.Generator().manual_seed(5)
    L = 8000
    batches = []
    for _ in range(2):
        batches.append((torch.randn(2, 1, L, generator=gen) * 0.2, torch.tensor([L, 5555]), torch.randn(2, 560, L // 256, generator=gen)))

    # ---- CPU reference loop: the statements of training_step on float64 leaves ----
    g_names = [k for k in gsd if k.startswith(("encoder.", "quantizer.", "decoder.", "quality_projection."))]
    og, od = opt([gsd[k] for k in g_names]), opt(list(dsd.values()))
    sg, sdl = sched(og), sched(od)
    ref_logs = []

    def avg(x, m):
        return (x * m).sum() / m.expand_as(x).sum()

    for audio, lens, noise in batches:
        _, gen_mel, gt = ref_cpu.vqgan_generator_loss(gsd, cfg, audio, lens, noise)
        mask = (torch.arange(gt.shape[2])[None, :] < (lens // 256)[:, None])[:, None, :].double()
        real = ref_cpu.discriminator_forward(dsd, "", gt)
        fake = ref_cpu.discriminator_forward(dsd, "", gen_mel.detach())
        dmask = F.interpolate(mask, size=(real.shape[2],), mode="nearest")
        loss_d = avg((real - 1) ** 2, dmask) + avg(fake ** 2, dmask)
        loss_d.backward()
        torch.nn.utils.clip_grad_norm_(list(dsd.values()), 1000.0)
        od.step(); od.zero_grad(); sdl.step()
        dist = (gen_mel - gt).abs()
        loss_mel = (avg(dist[:, :40], mask) * 0.6 + avg(dist[:, 40:70], mask) * 0.3 + avg(dist[:, 70:], mask) * 0.1) * 0.5 + avg(dist, mask) * 0.5
        loss_adv = avg((ref_cpu.discriminator_forward(dsd, "", gen_mel) - 1) ** 2, dmask)
        loss = loss_mel + loss_adv
        loss.backward()
        torch.nn.utils.clip_grad_norm_([gsd[k] for k in g_names], 1000.0)
        og.step(); og.zero_grad(); sg.step()
        ref_logs.append((loss_d.item(), loss_mel.item(), loss_adv.item()))

    # ---- native ----
    codec = codec.to(dev)
    for i, (audio, lens, noise) in enumerate(batches):
        logs = codec.training_step({"audios": audio.to(dev), "audio_lengths": lens.to(dev)}, i, noise=noise.to(dev))
        rd, rm, ra = ref_logs[i]
        # north_star: losses within 1e-4 relative in fp32
        assert abs(logs["train/discriminator/loss"] - rd) < 1e-4 * abs(rd), (i, logs, ref_logs[i])
        assert abs(logs["train/generator/loss_mel"] - rm) < 1e-4 * abs(rm), (i, logs, ref_logs[i])
        assert abs(logs["train/generator/loss_adv"] - ra) < 1e-4 * abs(ra), (i, logs, ref_logs[i])
    after = cpu_sd(codec)
    # AdamW normalises every element's step to ~lr, so elements whose gradient is near zero turn rounding noise into O(lr) differences:
    # the bar on parameters is a fraction of the accumulated step (2 x lr = 4e-3 absolute), not the 1e-4 of a forward pass
    for k in g_names:
        assert rel_err(after[k], gsd[k].detach()) < 3e-3, (k, rel_err(after[k], gsd[k].detach()))
        assert (after[k].double() - gsd[k].detach()).abs().mean() < 2e-5, k
    for k, v in dsd.items():
        assert rel_err(after["discriminator." + k], v.detach()) < 3e-3, (k, rel_err(after["discriminator." + k], v.detach()))
        assert (after["discriminator." + k].double() - v.detach()).abs().mean() < 2e-5, k


def test_f16_split_debug_range_check_is_loud(dev):
    """DMEL_DEBUG_F16_RANGE=1 (include/dmel_hip.h, DMEL_PRECISION_FP32_F16X2): a launch whose input leaves the fp16 split's domain
    (|x| >= 2^6 * 65504) fails with an error instead of producing inf; inside the domain nothing changes.  The switch is read once per
    process, hence the child process."""
    import subprocess, sys, os, textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import ctypes as C, torch
        from dmel_codec_amd import _lib
        L = _lib.lib()
        h = C.c_void_p()
        w = torch.randn(32, 32, 3) * 0.1
        _lib.check(L.dmel_conv_create(C.byref(h), w.data_ptr(), None, 32, 32, 3, 1))
        _lib.check(L.dmel_conv_set_precision(h, 3))
        x = torch.randn(2, 32, 200, device="cuda")
        y = torch.empty_like(x)
        _lib.check(L.dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), 2, 200, _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        x[1, 7, 100] = 5.0e6
        try:
            _lib.check(L.dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), 2, 200, _lib.stream_ptr()))
            print("NOT-REFUSED")
        except RuntimeError as e:
            print("REFUSED" if "outside the documented domain" in str(e) else "OTHER " + str(e))
    """)
    env = dict(os.environ, DMEL_DEBUG_F16_RANGE="1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "REFUSED", out.stdout + out.stderr[-1000:]
