"""CPU-side checks of the boundary: the C-ABI library builds, loads without a GPU, exports every symbol
include/dmel_hip.h declares, fails loudly (no fallback), and its host-side logic matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import ref_cpu


@pytest.fixture(scope="module")
def L():
    from dmel_codec_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from dmel_codec_amd.build import build
        build(verbose=False)
    return _lib.lib()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dmel_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dmel_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from dmel_codec_amd import _lib
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dmel_hip.h but not exported by libdmel_hip.so"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert L.dmel_abi_version() == 2


def test_errors_are_loud_not_fallbacks(L):
    from dmel_codec_amd import _lib
    h = C.c_void_p()
    rc = L.dmel_stft_plan_create(C.byref(h), 44100, 4096, 4096, 1024, 128, 0.0, 0.0, None)
    assert rc == -2 and b"n_fft" in L.dmel_last_error()
    with pytest.raises(RuntimeError, match="n_fft"):
        _lib.check(rc, "stft_plan_create")
    lv = (C.c_int * 3)(7, 5, 5)
    fs = (C.c_int * 2)(2, 3)
    assert L.dmel_quantizer_create(C.byref(h), 700, 10, lv, 3, fs, 2, 1) == -2
    assert L.dmel_quantizer_create(C.byref(h), 701, 10, lv, 3, fs, 2, 1) == -1
    # CPU tensors are refused by the mirror modules (there is no CPU path)
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    with pytest.raises(RuntimeError, match="GPU"):
        LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=100)(torch.zeros(1, 4000))


def test_missing_weights_are_reported(L):
    h = C.c_void_p()
    assert L.dmel_wavenet_create(C.byref(h), 10, 0, 70, 2, 4, 0) == 0
    rc = L.dmel_wavenet_finalize(h)
    assert rc == -3 and b"missing state-dict tensor 'input_projection.conv." in L.dmel_last_error()
    w = torch.zeros(70, 11, 1)
    shape = (C.c_int64 * 3)(70, 11, 1)
    assert L.dmel_wavenet_set_tensor(h, b"input_projection.conv.weight", w.data_ptr(), shape, 3) == 0
    bshape = (C.c_int64 * 1)(70)
    assert L.dmel_wavenet_set_tensor(h, b"input_projection.conv.bias", w.data_ptr(), bshape, 1) == 0
    assert L.dmel_wavenet_finalize(h) == -3 and b"shape" in L.dmel_last_error()
    L.dmel_wavenet_destroy(h)


def test_mel_basis_matches_oracle(L):
    for sr, n_mels, fmax in ((24000, 100, 12000.0), (16000, 80, None), (24000, 80, None), (44100, 128, None)):
        out = torch.empty(n_mels, 513)
        assert L.dmel_mel_basis_host(sr, 1024, n_mels, 0.0, fmax or 0.0, out.data_ptr()) == 0
        ref = ref_cpu.slaney_mel_basis(sr, 1024, n_mels, 0.0, fmax)
        assert np.allclose(out.numpy(), ref, rtol=0, atol=1e-9)
        assert np.array_equal(out.numpy() != 0, ref != 0)


def test_mirror_state_dict_layout():
    """Key names are part of the drop-in contract."""
    from dmel_codec_amd.configs import build_codec
    m = build_codec(n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=2)
    sd = m.state_dict()
    for k, shape in {
        "encoder.input_projection.conv.weight": (70, 10, 1),
        "encoder.residual_layers.1.conv_layer.conv.weight": (140, 70, 3),
        "encoder.residual_layers.0.output_projection.conv.bias": (140,),
        "encoder.skip_projection.conv.weight": (70, 70, 1),
        "quantizer.residual_fsq.rvqs.7.project_in.weight": (3, 70),
        "quantizer.residual_fsq.rvqs.0.project_out.bias": (70,),
        "quantizer.downsample.0.0.weight": (70, 70, 2),
        "quantizer.downsample.1.1.dwconv.weight": (70, 1, 7),
        "quantizer.downsample.0.1.pwconv1.weight": (280, 70),
        "quantizer.upsample.1.0.weight": (70, 70, 2),
        "quantizer.upsample.0.1.gamma": (70,),
        "decoder.residual_layers.0.condition_projection.conv.weight": (1120, 560, 1),
        "decoder.residual_layers.1.diffusion_projection.linear.weight": (560, 560),
        "decoder.output_projection.conv.weight": (80, 560, 1),
        "quality_projection.weight": (560, 1),
        "vocoder.conv_pre.weight_g": (512, 1, 1),
        "vocoder.conv_pre.weight_v": (512, 80, 7),
        "vocoder.ups.0.0.weight_v": (512, 256, 16),
        "vocoder.resblocks.0.convs1.2.weight_v": (256, 256, 3),
        "vocoder.resblocks.11.activations.5.act.beta": (32,),
        "vocoder.resblocks.3.activations.0.upsample.filter": (1, 1, 12),
        "vocoder.resblocks.3.activations.0.downsample.lowpass.filter": (1, 1, 12),
        "vocoder.activation_post.act.alpha": (32,),
        "vocoder.conv_post.weight_v": (1, 32, 7),
    }.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shape, (k, tuple(sd[k].shape))
    assert "decoder.input_projection.conv.weight" not in sd      # 560 == 560: wavenet.py:152-156


@pytest.mark.skipif(not os.path.isdir("/root/reference/dmel_codec"), reason="reference only exists in the build container")
def test_mirror_state_dicts_equal_reference_modules():
    import subprocess
    import sys
    code = r"""
import sys, warnings
warnings.filterwarnings('ignore')
sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')   # the reference's dmel_codec must win over the alias package
sys.dont_write_bytecode = True
from dmel_codec.models.modules.wavenet import WaveNet as RW
from dmel_codec.models.modules.bigvgan.bigvgan import BigVGAN as RB
from dmel_codec.models.modules.bigvgan.env import AttrDict as RA
from dmel_codec.models.modules.firefly import ConvNeXtBlock as RC
from dmel_codec_amd.models.modules.wavenet import WaveNet
from dmel_codec_amd.models.modules.bigvgan.bigvgan import BigVGAN
from dmel_codec_amd.models.modules.firefly import ConvNeXtBlock
from dmel_codec_amd.configs import bigvgan_h, BIGVGAN
def same(a, b):
    sa = {k: tuple(v.shape) for k, v in a.state_dict().items()}
    sb = {k: tuple(v.shape) for k, v in b.state_dict().items()}
    assert sa == sb, (set(sa) ^ set(sb))
same(RW(10, None, 70, 20, 4), WaveNet(10, None, 70, 20, 4))
same(RW(64, 20, 64, 3, 4, False, 64), WaveNet(64, 20, 64, 3, 4, False, 64))
same(RC(70), ConvNeXtBlock(70))
for name in ('base_24k_100band',):
    r, m = RB(RA(dict(BIGVGAN[name]))), BigVGAN(bigvgan_h(name))
    same(r, m)
    r.remove_weight_norm(); m.remove_weight_norm()
    same(r, m)
print('ok')
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_yaml_config_builds_the_codec():
    """The reference's config schema (`_target_`, `_partial_`, `${a.b}`, `defaults[1:]` merge of train_codec.py:12-23)
    builds the MI355X codec; `_target_` strings keep the REFERENCE's dotted paths."""
    import functools
    from dmel_codec_amd import config_loader
    from dmel_codec_amd.models.codec_lit_modules import VQGAN
    path = os.path.join(ROOT, "dmel_codec_amd", "config", "codec", "dMel_mi355x.yaml")
    cfg = config_loader.load_config(path)
    assert cfg["concat_channels_dim"] == 700 and cfg["model"]["decoder"]["residual_channels"] == 700
    assert cfg["model"]["vocoder"]["h_path"].endswith("base_24k_100band.json")
    m = config_loader.build_codec_from_config(path, overrides={"model": {"encoder": {"residual_layers": 2},
                                                                         "decoder": {"residual_layers": 2}}},
                                              load_vocoder_ckpt=False)
    assert isinstance(m, VQGAN) and m.dmel_groups == 10 and m.vocoder.h.upsample_rates == [8, 8, 2, 2]
    assert isinstance(m.optimizer_builder, functools.partial) and m.optimizer_builder.keywords["lr"] == 1e-4
    lam = m.lr_scheduler_builder.keywords["lr_lambda"]
    assert abs(lam(50) - 0.5) < 1e-12
    with pytest.raises(ValueError, match="concat_channels_dim"):
        config_loader.resolve({"concat_channels_dim": "???", "model": {"x": "${concat_channels_dim}"}})


def test_alias_package_serves_reference_paths():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from dmel_codec.models.modules.wavenet import WaveNet\n"
            "from dmel_codec.utils.spectrogram import LogMelSpectrogram\n"
            "from dmel_codec.models.lit_modules import VQGAN\n"
            "from dmel_codec.models.modules.bigvgan.bigvgan import BigVGAN\n"
            "from dmel_codec.models.modules.bigvgan.alias_free_activation.torch.act import Activation1d\n"
            "import dmel_codec_amd.models.modules.wavenet as w\n"
            "assert WaveNet is w.WaveNet\nprint('ok')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]


@pytest.mark.skipif(not os.path.isfile("/root/reference/dmel_codec/config/codec/dMel_example.yaml"),
                    reason="reference only exists in the build container")
def test_reference_yaml_instantiates():
    """The reference's own example config (read as data, in place) builds the codec once the author-local vocoder
    paths are overridden."""
    from dmel_codec_amd import config_loader
    h = os.path.join(ROOT, "dmel_codec_amd", "config", "bigvgan", "base_24k_100band.json")
    m = config_loader.build_codec_from_config("/root/reference/dmel_codec/config/codec/dMel_example.yaml",
                                              overrides={"model": {"vocoder": {"h_path": h, "ckpt_path": None}}},
                                              load_vocoder_ckpt=False)
    assert len(m.encoder.residual_layers) == 8 and m.quantizer.levels == [8, 6] and m.dmel_groups == 10
    assert m.decoder.condition_channels == 700 and m.encode_mel_transform.n_mels == 100


def test_precision_api_validation():
    """set_precision is host-side state until a handle exists: argument checking needs no GPU."""
    import pytest
    from dmel_codec_amd.configs import build_codec
    codec = build_codec(n_mels=80, dmel_groups=8, encoder_layers=1, decoder_layers=1, vocoder=None)
    for ok in ("fp32", "bf16", "fp32_mfma", torch.bfloat16, torch.float32, 0, 1, 2):
        codec.decoder.set_precision(ok)
    with pytest.raises(ValueError):
        codec.decoder.set_precision("fp16")
    with pytest.raises(NotImplementedError):
        codec.quantizer.set_precision("bf16")          # ids are the interchange format: the quantizer has no bf16 mode
    codec.quantizer.set_precision("fp32")
    codec.set_decode_precision("bf16")
    assert codec.decoder._precision == 1 and codec.encoder._precision == 0


def test_precision_entry_points_reject_bad_arguments(L):
    assert L.dmel_conv_set_precision(None, 0) < 0 and b"PRECISION" in L.dmel_last_error()
    assert L.dmel_wavenet_set_precision(None, 1) < 0
    assert L.dmel_bigvgan_set_precision(None, 1) < 0


def test_handle_cache_is_tied_to_the_tensor_not_to_its_address():
    """torch_ops caches packed-weight handles per weight tensor.  The key must be the tensor OBJECT and its version, never data_ptr():
    the allocator hands a freed weight's address to the next tensor of that shape, and a fresh tensor starts at version 0 again."""
    import gc
    from dmel_codec_amd.torch_ops import _TensorKeyedCache
    destroyed = []
    cache = _TensorKeyedCache(3, destroy=destroyed.append)
    w, b = torch.zeros(4, 4, 3), torch.zeros(4)
    assert cache.get(w, 1, b) is None
    cache.put(w, 1, "h1", b)
    assert cache.get(w, 1, b) == "h1"
    assert cache.get(w, 2, b) is None                                   # another dilation: its own entry
    assert cache.get(w, 1, None, any_other=True) == "h1"                # backward does not care which bias the handle was built with
    assert cache.get(w, 1, torch.zeros(4)) is None and destroyed == ["h1"]    # a different bias object: stale, destroyed, rebuilt by the caller
    cache.put(w, 1, "h2", b)
    w.add_(1.0)                                                         # in-place update (an optimiser step): version moves
    assert cache.get(w, 1, b) is None and destroyed == ["h1", "h2"]
    cache.put(w, 1, "h3", b)
    b.mul_(2.0)
    assert cache.get(w, 1, b) is None and destroyed[-1] == "h3"
    # the tensor dies: its handle is destroyed, and a NEW tensor (same shape, version 0, quite possibly the same address / id) misses
    cache.put(w, 1, "h4", b)
    del w
    gc.collect()
    assert destroyed[-1] == "h4" and len(cache.entries) == 0
    w2 = torch.zeros(4, 4, 3)
    assert cache.get(w2, 1, b) is None
    # least recently used goes first, one at a time
    ts = [torch.zeros(2) for _ in range(4)]
    for i, t in enumerate(ts[:3]):
        cache.put(t, 0, f"t{i}")
    assert cache.get(ts[0], 0) == "t0"                                  # touch 0: 1 is now the oldest
    cache.put(ts[3], 0, "t3")
    assert destroyed[-1] == "t1" and cache.get(ts[0], 0) == "t0" and cache.get(ts[2], 0) == "t2" and cache.get(ts[3], 0) == "t3"


def test_copies_of_a_mirror_never_share_its_native_handle():
    """copy.deepcopy (what dmel_codec_amd.pipeline.CodecLanes makes its lanes with) and pickling of a mirror module: the copy carries no
    handle, no plan and no workspace of the original -- a handle is an address inside the library; copied by value, the copy's first
    native() would free the original's."""
    import copy
    import pickle
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    from dmel_codec_amd.utils.spectrogram import LogMelSpectrogram
    m = WaveNet(input_channels=10, output_channels=10, residual_channels=16, residual_layers=2, dilation_cycle=2)
    m._handle, m._handle_versions = 0xDEAD0000, ("stale",)      # pretend a handle exists (never dereferenced: reset below)
    m._ws.buf = torch.zeros(4)
    try:
        for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
            assert c._handle is None and c._handle_versions is None and c._ws is not m._ws and c._ws.buf is None
            assert len(c._train_calls) == 0 and c._grad_sink is None
            assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
        assert m._handle == 0xDEAD0000
    finally:
        m._handle = None      # or __del__ would hand the fake address to the library
    s = LogMelSpectrogram(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=80)
    s.spectrogram._plan = 0xBEEF0000
    try:
        assert copy.deepcopy(s).spectrogram._plan is None
    finally:
        s.spectrogram._plan = None


def test_lanes_refuse_a_cpu_codec():
    """dmel_codec_amd.pipeline.CodecLanes has no CPU form either: a codec that is not on a GPU raises (it would otherwise copy the codec
    and fail later, inside the first native call)."""
    from dmel_codec_amd.configs import build_codec
    from dmel_codec_amd.pipeline import CodecLanes
    codec = build_codec(n_mels=80, dmel_groups=8, encoder_layers=1, decoder_layers=1, vocoder=None)
    with pytest.raises(RuntimeError, match="CUDA"):
        CodecLanes(codec, 2)
    with pytest.raises(ValueError):
        CodecLanes(codec, 0)
