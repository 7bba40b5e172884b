"""The CPU oracle (oracle/ref_cpu.py) against the fixtures generated from the reference's own
modules (oracle/gen_golden.py).  CPU only; runs in the build container and on the GPU box."""
import numpy as np
import torch

from oracle import ref_cpu
from conftest import rel_err

TOL = 2e-6   # same ATen-CPU calls as the reference: differences are thread-count/blocking noise


def test_wavenet_encoder(golden):
    g = golden("wavenet_enc")
    y = ref_cpu.wavenet_forward(g.sd, "", g.ins["x"], g.meta["n_layers"], g.meta["dilation_cycle"])
    assert rel_err(y, g.outs["y"]) < TOL


def test_wavenet_decoder_conditioned(golden):
    g = golden("wavenet_dec")
    y = ref_cpu.wavenet_forward(g.sd, "", g.ins["x"], g.meta["n_layers"], g.meta["dilation_cycle"],
                                condition=g.ins["cond"])
    assert y.shape == g.outs["y"].shape
    assert rel_err(y, g.outs["y"]) < TOL


def test_convnext(golden):
    g = golden("convnext")
    assert rel_err(ref_cpu.convnext_block(g.sd, "", g.ins["x"]), g.outs["y"]) < TOL


def test_activation1d(golden):
    for kind in ("snakebeta", "snake"):
        g = golden("activation1d_" + kind)
        for t in g.meta["lengths"]:
            y = ref_cpu.activation1d(g.ins[f"x{t}"], g.sd["act.alpha"], g.sd.get("act.beta", g.sd["act.alpha"]),
                                     g.sd["upsample.filter"], g.sd["downsample.lowpass.filter"],
                                     logscale=True, snake=(kind == "snake"))
            assert rel_err(y, g.outs[f"y{t}"]) < TOL, (kind, t)


def test_aa_filter_taps(golden):
    g = golden("aa_filter")
    assert torch.equal(ref_cpu.aa_filter12(), g.outs["taps12"])
    # SURVEY App. A.5 prints the taps
    assert abs(float(g.outs["taps12"][0, 0, 5]) - 0.4432097971) < 1e-9


def test_ampblock1(golden):
    g = golden("ampblock1")
    y = ref_cpu.ampblock1(g.sd, "", g.ins["x"], g.meta["k"], g.meta["dilations"])
    assert rel_err(y, g.outs["y"]) < TOL


def test_bigvgan_tiny(golden):
    for name in ("bigvgan_tiny", "bigvgan_tiny_snake_nowm"):
        g = golden(name)
        y = ref_cpu.bigvgan_forward(g.sd, g.meta["h"], g.ins["mel"])
        assert y.shape == g.outs["audio"].shape
        assert rel_err(y, g.outs["audio"]) < 5e-6, name


def test_mel_basis_selfchecks(golden):
    """Parity of the mel basis is UNPINNED by the reference (librosa absent): formula self-checks."""
    g = golden("mel_basis")
    for tag, (sr, n_mels, fmax) in {"24k100": (24000, 100, 12000.0), "16k80": (16000, 80, None)}.items():
        m = ref_cpu.slaney_mel_basis(sr, 1024, n_mels, 0.0, fmax)
        assert m.dtype == np.float32 and m.shape == (n_mels, 513)
        assert np.array_equal(m, g.outs[tag].numpy())
        assert m[0, 0] == 0.0 and (m >= 0).all()
        # every filter is a single triangle: one contiguous support, one peak
        for row in m:
            nz = np.nonzero(row)[0]
            assert len(nz) > 0 and nz[-1] - nz[0] + 1 == len(nz)
        # Slaney area normalisation: integral of each triangle over Hz is ~1
        area = m.sum(axis=1) * (sr / 1024)
        assert np.allclose(area[10:], 1.0, atol=0.12)
        # linear region below 1 kHz: centre spacing is constant
        peaks = m.argmax(axis=1) * sr / 1024
        low = peaks[peaks < 900]
        assert np.ptp(np.diff(low)) <= 2 * sr / 1024


def test_stft_logmel_fixture(golden):
    g = golden("stft_logmel")
    for tag, (sr, n_mels, fmax) in g.meta["cases"].items():
        y = ref_cpu.stft_logmel(g.ins[tag], sr, 1024, 1024, 256, n_mels, 0.0, fmax)
        assert y.shape == (2, n_mels, sr // 256)
        assert rel_err(y, g.outs[tag]) < TOL


def test_stft_against_direct_dft():
    """torch.stft is the reference's own call; check the oracle's framing/padding/window
    against a float64 direct DFT."""
    torch.manual_seed(3)
    L, n_fft, hop = 2000, 256, 64
    y = torch.randn(1, L)
    mag = ref_cpu.stft_logmel(y, 16000, n_fft, n_fft, hop, 20, return_linear=True)[0].double()
    pad = (n_fft - hop) // 2
    yp = torch.nn.functional.pad(y.double().unsqueeze(1), (pad, pad), mode="reflect")[0, 0]
    n = torch.arange(n_fft, dtype=torch.float64)
    w = 0.5 - 0.5 * torch.cos(2 * torch.pi * n / n_fft)
    T = L // hop
    assert mag.shape == (n_fft // 2 + 1, T)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64)
    e = torch.exp(-2j * torch.pi * k[:, None] * n[None, :] / n_fft)
    for t in (0, 1, T // 2, T - 1):
        fr = yp[t * hop:t * hop + n_fft] * w
        X = (e * fr[None, :].to(torch.complex128)).sum(1)
        ref = torch.sqrt(X.real ** 2 + X.imag ** 2 + 1e-9)
        assert torch.allclose(mag[:, t], ref, rtol=1e-4, atol=1e-4)


def test_fsq_fixture_and_invariants(golden):
    for levels in ([7, 5, 5], [8, 6]):
        g = golden("fsq_" + "x".join(map(str, levels)))
        n_codes = int(np.prod(levels))
        for pb in (1, 0):
            ids, pre = ref_cpu.grouped_fsq_encode(g.sd, "", g.ins["z"], 1, levels, prebound=bool(pb),
                                                  return_prequant=True)
            assert torch.equal(ids, g.outs[f"ids_prebound{pb}"])
            assert ids.dtype == torch.int32 and int(ids.min()) >= 0 and int(ids.max()) < n_codes
        # decode(encode) round trip on the implicit codebook: ids -> codes -> ids
        codes = ref_cpu.fsq_codes_from_indices(torch.arange(n_codes), levels)
        hw = torch.tensor(levels) // 2
        back = ref_cpu.fsq_indices_from_prequant(codes * hw, levels)
        assert torch.equal(back, torch.arange(n_codes, dtype=torch.int32))
        deq = ref_cpu.grouped_fsq_decode(g.sd, "", g.ins["all_ids"], levels)
        assert torch.equal(deq, g.outs["dequant"])
    # reference-side invariants that pin the codebook size: 175 ids, silence frame < 175
    assert int(np.prod([7, 5, 5])) == 175
    assert max([0, 0, 29, 174, 0, 6, 0, 146, 146, 6]) < 175


def test_schedule_fixture(golden):
    """utils/schedule.py:4-25: the product's schedule function against values produced by the reference's."""
    from dmel_codec_amd.utils.schedule import get_cosine_schedule_with_warmup_lr_lambda as lr_lambda
    g = golden("schedule")
    vals = torch.tensor([lr_lambda(int(s), num_warmup_steps=100, num_training_steps=1000000, final_lr_ratio=0.05)
                         for s in g.ins["steps"]])
    assert torch.allclose(vals, g.outs["lr"], atol=1e-7)
    assert lr_lambda(10, num_warmup_steps=0.1, num_training_steps=1000) == 0.1      # fractional warm-up


def _oracle_grads(y, dy, leaves):
    (y * dy).sum().backward()
    return {k: v.grad for k, v in leaves.items()}


def test_training_gradients_match_reference(golden):
    """The GPU training tests use autograd through the oracle (in float64) as their truth; these fixtures hold what autograd through
    the REFERENCE's own modules produced (oracle/gen_golden_grads.py), so the oracle's backward is pinned to the reference too."""
    g = golden("train_grads_wavenet")
    sd = {k: v.clone().requires_grad_() for k, v in g.sd.items()}
    x, c = g.ins["x"].clone().requires_grad_(), g.ins["cond"].clone().requires_grad_()
    y = ref_cpu.wavenet_forward(sd, "", x, g.meta["n_layers"], g.meta["dilation_cycle"], c)
    assert rel_err(y, g.outs["y"]) < 5e-6
    (y * g.ins["dy"]).sum().backward()
    assert rel_err(x.grad, g.outs["d_x"]) < 2e-5 and rel_err(c.grad, g.outs["d_cond"]) < 2e-5
    n = 0
    for k, ref in g.outs.items():
        if k.startswith("g/"):
            assert rel_err(sd[k[2:]].grad, ref) < 5e-5, k
            n += 1
    assert n == len([k for k in g.sd if "diffusion_projection" not in k])
    assert all(sd[k].grad is None for k in g.sd if "diffusion_projection" in k)      # dead parameters, in the reference too

    g = golden("train_grads_convnext")
    sd = {k: v.clone().requires_grad_() for k, v in g.sd.items()}
    x = g.ins["x"].clone().requires_grad_()
    y = ref_cpu.convnext_block(sd, "", x)
    (y * g.ins["dy"]).sum().backward()
    assert rel_err(y, g.outs["y"]) < 5e-6 and rel_err(x.grad, g.outs["d_x"]) < 2e-5
    for k, ref in g.outs.items():
        if k.startswith("g/"):
            assert rel_err(sd[k[2:]].grad, ref) < 5e-5, k

    g = golden("train_grads_activation1d")
    a, b = g.sd["act.alpha"].clone().requires_grad_(), g.sd["act.beta"].clone().requires_grad_()
    x = g.ins["x"].clone().requires_grad_()
    y = ref_cpu.activation1d(x, a, b, g.sd["upsample.filter"], g.sd["downsample.lowpass.filter"], logscale=True)
    (y * g.ins["dy"]).sum().backward()
    assert rel_err(y, g.outs["y"]) < 5e-6 and rel_err(x.grad, g.outs["d_x"]) < 2e-5
    assert rel_err(a.grad, g.outs["g/act.alpha"]) < 5e-5 and rel_err(b.grad, g.outs["g/act.beta"]) < 5e-5


def test_discriminator_oracle_matches_reference(golden):
    """Logits and autograd gradients of the reference's Discriminator class on seeded weights (fixture: seed + data) against the
    oracle's restatement."""
    g = golden("train_grads_discriminator")
    sd = {k: v.clone().requires_grad_() for k, v in ref_cpu.seeded_discriminator_sd(g.meta["seed"]).items()}
    x = g.ins["x"].clone().requires_grad_()
    y = ref_cpu.discriminator_forward(sd, "", x)
    assert rel_err(y, g.outs["y"]) < 1e-5
    (y * g.ins["dy"]).sum().backward()
    assert rel_err(x.grad, g.outs["d_x"]) < 5e-5
    for k, ref in g.outs.items():
        if k.startswith("g/"):
            assert rel_err(sd[k[2:]].grad, ref) < 1e-4, k
        elif k.startswith("proj/"):
            p = sd[k[5:]]
            r = torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel()))
            assert abs(float((p.grad * r).sum()) - float(ref)) < 1e-3 * max(1.0, abs(float(ref))), k


def test_mel_basis_matches_an_independent_slaney_implementation():
    """librosa (the reference's source of the mel basis, utils/spectrogram.py:45-52) is absent, and the reference holds no fixture
    for it; `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` is an independent implementation of the
    same published definition (it is what HF feature extractors use in place of librosa.filters.mel) and IS installed here: the
    oracle's restatement and the product's C++ restatement (tests/test_cpu_abi.py::test_mel_basis_matches_oracle holds those two
    together) agree with it to float32 rounding on every configuration of the codec."""
    import pytest
    audio_utils = pytest.importorskip("transformers.audio_utils")
    for sr, n_fft, n_mels, fmax in [(24000, 1024, 100, 12000.0), (16000, 1024, 80, None), (24000, 1024, 80, None), (44100, 2048, 128, None)]:
        ours = np.asarray(ref_cpu.slaney_mel_basis(sr, n_fft, n_mels, 0.0, fmax))
        theirs = audio_utils.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=0.0,
                                             max_frequency=fmax if fmax is not None else sr / 2, sampling_rate=sr, norm="slaney",
                                             mel_scale="slaney").T
        assert ours.shape == theirs.shape == (n_mels, n_fft // 2 + 1)
        assert np.abs(ours - theirs).max() < 1e-8 and np.abs(ours - theirs).max() / np.abs(theirs).max() < 2e-7, (sr, n_mels)


def test_strict_fsq_restatement():
    """Strict encode (project_in + bound in float64, rounded once): equals the fp32 restatement up to fp32 rounding noise, is
    invariant to the summation order of the Linear (a permutation of the input channels changes nothing, while the fp32 path moves
    in the last bits), and yields the same ids away from rounding boundaries."""
    g = torch.Generator().manual_seed(3)
    levels = [7, 5, 5]
    sd = {"project_in.weight": torch.randn(3, 70, generator=g) * 0.4, "project_in.bias": torch.randn(3, generator=g) * 0.1}
    x = torch.randn(2000, 70, generator=g)
    for prebound in (True, False):
        a = ref_cpu.fsq_prequant(sd, "", x, levels, prebound)
        s = ref_cpu.fsq_prequant(sd, "", x, levels, prebound, strict=True)
        assert s.dtype == torch.float32 and (a - s).abs().max() < 5e-5
        perm = torch.randperm(70, generator=g)
        sd_p = {"project_in.weight": sd["project_in.weight"][:, perm], "project_in.bias": sd["project_in.bias"]}
        assert torch.equal(ref_cpu.fsq_prequant(sd_p, "", x[:, perm], levels, prebound, strict=True), s)
        far = ((a - a.floor() - 0.5).abs() > 1e-4).all(-1)
        assert torch.equal(ref_cpu.fsq_indices_from_prequant(a, levels)[far], ref_cpu.fsq_indices_from_prequant(s, levels)[far])


def test_ampblock2_and_bigvgan_resblock2(golden):
    """AMPBlock2 (`resblock: "2"`, bigvgan.py:150-241): the oracle against outputs of the reference's own classes."""
    g = golden("ampblock2")
    assert rel_err(ref_cpu.ampblock2(g.sd, "", g.ins["x"], g.meta["k"], g.meta["dilations"]), g.outs["y"]) < TOL
    g = golden("bigvgan_tiny_ampblock2")
    y = ref_cpu.bigvgan_forward(g.sd, g.meta["h"], g.ins["mel"])
    assert y.shape == g.outs["audio"].shape and rel_err(y, g.outs["audio"]) < 5e-6
    # the mirror's state-dict layout equals the reference's (keys and shapes)
    from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
    from dmel_codec_amd.models.modules.bigvgan.env import AttrDict
    m = BigVGAN(AttrDict(dict(g.meta["h"])))
    sd = m.state_dict()
    assert set(sd) == set(g.sd) and all(tuple(sd[k].shape) == tuple(g.sd[k].shape) for k in sd)
