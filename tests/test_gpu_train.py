"""GPU tests of the training entry point and its exchange step: train_codec.py (config -> trainer -> checkpoints -> resume), the
per-block gradient hand-over of the native WaveNet backward (the hook the RCCL overlap rides on), checkpoint wire formats."""
import os

import pytest
import torch

from test_gpu_parity import cpu_sd, make_codec, randomise  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


class _Recorder:
    """Stands in for ddp.GradReducer on a single rank: records what the native backward hands over, and when."""

    def __init__(self):
        self.regions = []

    def submit(self, flat):
        # the hand-over happens on the host while the GPU is still far behind: nothing may have been synchronised
        self.regions.append((flat.data_ptr(), flat.numel()))


@pytest.mark.parametrize("cfg", [dict(input_channels=10, residual_channels=32, residual_layers=5),
                                 dict(input_channels=64, output_channels=20, residual_channels=64, residual_layers=4, condition_channels=64)])
def test_wavenet_backward_hands_over_gradients_block_by_block(dev, cfg):
    """dmel_wavenet_backward_hooked: tail first, then residual_layers L-1 ... 0, then input_projection; disjoint regions that cover the
    whole flat buffer; parameter .grad are views of that buffer (the all-reduce is in place); values equal the plain autograd path."""
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    torch.manual_seed(3)
    m = WaveNet(dilation_cycle=4, **cfg)
    randomise(m, 11)
    m = m.to(dev)
    N, T = 3, 50
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, cfg["input_channels"], T, generator=g).to(dev)
    c = torch.randn(N, cfg["condition_channels"], T, generator=g).to(dev) if cfg.get("condition_channels") else None
    m(x, condition=c).square().sum().backward()
    plain = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    rec = _Recorder()
    m._grad_sink = rec
    m(x, condition=c).square().sum().backward()
    m._grad_sink = None
    L = cfg["residual_layers"]
    trained = m._trained_parameters()
    has_in = m.input_projection is not None
    assert len(rec.regions) == L + 1 + int(has_in)
    base = min(p for p, _ in rec.regions)
    spans = [(p - base, n * 4) for p, n in rec.regions]
    total = sum(p.numel() for _, p in trained) * 4
    assert sorted(spans)[0][0] == 0 and sum(n for _, n in spans) == total
    ordered = sorted(spans)
    assert all(a[0] + a[1] == b[0] for a, b in zip(ordered, ordered[1:]))          # disjoint, contiguous cover
    # order of hand-over: the tail lives at the END of the buffer, blocks follow from the last to the first, the head is at offset 0
    offs = [o for o, _ in spans]
    assert offs[0] == max(offs)
    assert offs[1:L + 1] == sorted(offs[1:L + 1], reverse=True)
    if has_in:
        assert offs[-1] == 0
    per_block = sum(p.numel() for k, p in trained if k.startswith("residual_layers.0.")) * 4
    assert all(n == per_block for _, n in spans[1:L + 1])
    for k, p in trained:
        assert p.grad is not None and base <= p.grad.data_ptr() < base + total, k   # views of the flat buffer
        assert torch.allclose(p.grad, plain[k], rtol=1e-5, atol=1e-5 * float(plain[k].abs().max())), k   # fp32 atomics: run-to-run rounding
    assert all(p.grad is None for k, p in m.named_parameters() if "diffusion_projection" in k)
    # gradients already present: nothing is streamed, the whole buffer (old + new) goes out once, after the add
    rec2 = _Recorder()
    m._grad_sink = rec2
    m(x, condition=c).square().sum().backward()
    m._grad_sink = None
    assert len(rec2.regions) == 1 and rec2.regions[0][1] * 4 == total
    for k, p in trained:
        assert torch.allclose(p.grad, 2 * plain[k], rtol=1e-5, atol=2e-5 * float(plain[k].abs().max())), k


def test_backward_after_optimizer_step_is_refused(dev):
    """An optimiser step between a training forward and its backward re-packs the weight images in place; like torch autograd for
    in-place modified saved tensors, the native backward must refuse instead of mixing new weights with old activations."""
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    m = WaveNet(input_channels=8, residual_channels=16, residual_layers=2).to(dev)
    x = torch.randn(2, 8, 20, device=dev)
    y = m(x)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01)
    m(x)                      # refreshes the handle's images from the changed parameters
    with pytest.raises(RuntimeError, match="parameters changed"):
        y.sum().backward()
    m.zero_grad()
    m(x).sum().backward()     # and the module is usable afterwards
    assert m.skip_projection.conv.weight.grad is not None


TINY = ["model.encoder.residual_layers=2", "model.decoder.residual_layers=2", "data.train_max_durations=3", "data.val_max_durations=2",
        "data.train_batches_per_epoch=4", "data.val_batches=1", "trainer.log_every_n_steps=1", "trainer.val_check_interval=3",
        "callbacks.model_checkpoint.every_n_train_steps=4", "tensorboard_logger.save_dir=/tmp/dmel_tb", "model.optimizer.lr=1e-3"]


def test_train_codec_entry_point_trains_checkpoints_and_resumes(dev, tmp_path):
    """`python train_codec.py` (train_codec.py:12-64 of the reference): config merge -> data / model / callbacks / logger / trainer ->
    fit.  Three batches through the trainer give exactly the losses of calling VQGAN.training_step on the same batches by hand (the
    parity-tested step is what the entry point runs); checkpoints are Lightning-layout without vocoder keys; a second run resumes."""
    from dmel_codec_amd import config_loader
    from dmel_codec_amd.train_codec import _parse_overrides, get_config, main, seed_everything
    ck = str(tmp_path / "ckpt")
    over = _parse_overrides(TINY + [f"codec_ckpt_dir={ck}", "trainer.max_steps=6"])
    cfg = get_config(None, over)
    trainer = main(cfg)
    assert trainer.global_step == 6 and trainer.batches_seen == 3 and len(trainer.history) == 3
    # the same three batches by hand
    seed_everything(cfg["seed"])
    dm = config_loader.instantiate(cfg["data"])
    model = config_loader.instantiate(cfg["model"], load_vocoder_ckpt=False).to(dev)
    for i, batch in zip(range(3), dm.train_dataloader()):
        # the decoder draws its noise from torch's global generator: seed both runs identically around every step
        logs = model.training_step({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}, i)
        for k in ("train/discriminator/loss", "train/generator/loss_mel", "train/generator/loss_adv"):
            assert abs(logs[k] - trainer.history[i][k]) <= 2e-3 * abs(logs[k]) + 1e-6, (i, k, logs[k], trainer.history[i][k])
    files = sorted(os.listdir(ck))
    assert "last.ckpt" in files, files
    ckpt = torch.load(os.path.join(ck, "last.ckpt"), map_location="cpu", weights_only=False)
    keys = list(ckpt["state_dict"])
    assert not any("vocoder" in k for k in keys)                                            # codec_lit_modules.py:114-119
    assert any(k.startswith("encoder.residual_layers.0.conv_layer.conv.weight") for k in keys)
    assert any(k.startswith("quantizer.residual_fsq.rvqs.0.project_in.weight") for k in keys)
    assert any(k.startswith("discriminator.blocks.0.parametrizations.weight.original0") for k in keys)
    assert "quality_projection.weight" in keys and ckpt["global_step"] == 6
    # resume: newest *.ckpt, counters restored, two more batches
    cfg2 = get_config(None, _parse_overrides(TINY + [f"codec_ckpt_dir={ck}", "trainer.max_steps=10"]))
    trainer2 = main(cfg2)
    assert trainer2.global_step == 10 and trainer2.history[0]["step"] == 8 and trainer2.batches_seen == 5


def test_checkpoint_wire_formats_round_trip(dev, tmp_path):
    """a Lightning-layout codec checkpoint ({"state_dict": ...}, vocoder stripped, loaded with strict=False as
    evaluation/initial_codec.py:43 does) plus a BigVGAN file ({"generator": ...}, codec_lit_modules.py:68-72) rebuild a codec whose token
    ids and waveform are EXACTLY those of the original."""
    import json
    from dmel_codec_amd import config_loader
    from dmel_codec_amd.configs import BIGVGAN
    h = dict(BIGVGAN["base_24k_100band"], upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=64, num_mels=100)
    h_path, voc_path, codec_path = str(tmp_path / "config.json"), str(tmp_path / "bigvgan_generator.pt"), str(tmp_path / "codec.ckpt")
    with open(h_path, "w") as f:
        json.dump(h, f)
    over = {"model": {"encoder": {"residual_layers": 2}, "decoder": {"residual_layers": 2}, "vocoder": {"h_path": h_path, "ckpt_path": None}}}
    path = os.path.join(os.path.dirname(config_loader.__file__), "config", "codec", "dMel_mi355x.yaml")
    torch.manual_seed(0)
    a = config_loader.build_codec_from_config(path, overrides=over, load_vocoder_ckpt=False)
    randomise(a.encoder, 1); randomise(a.quantizer, 2, scale=1.5); randomise(a.decoder, 3); randomise(a.vocoder, 4, scale=0.7)
    # files, as the reference writes them
    checkpoint = {"state_dict": {k: v.detach().clone() for k, v in a.state_dict().items()}}
    a.on_save_checkpoint(checkpoint)
    assert not any("vocoder" in k for k in checkpoint["state_dict"])
    torch.save(checkpoint, codec_path)
    torch.save({"generator": a.vocoder.state_dict()}, voc_path)
    a = a.to(dev)
    g = torch.Generator().manual_seed(9)
    audio = (torch.randn(2, 1, 24000, generator=g) * 0.2).to(dev)
    lens = torch.tensor([24000, 17000], device=dev)
    ids_a, il_a = a.encode(audio, lens)
    noise = torch.randn(2, 700, ids_a.shape[2] * 4, generator=g).to(dev)
    wav_a, mel_a = a.decode(ids_a, il_a, return_audios=True, noise=noise)
    # rebuild from the files: the vocoder through its ckpt_path (VQGAN.__init__, codec_lit_modules.py:66-72), the rest strict=False
    torch.manual_seed(12345)
    over["model"]["vocoder"]["ckpt_path"] = voc_path
    b = config_loader.build_codec_from_config(path, overrides=over, load_vocoder_ckpt=True)
    assert b.vocoder is not None and b.decoder is not None
    missing, unexpected = b.load_state_dict(torch.load(codec_path, map_location="cpu")["state_dict"], strict=False)
    assert not unexpected and all(k.startswith("vocoder.") for k in missing)
    b = b.to(dev)
    ids_b, il_b = b.encode(audio, lens)
    wav_b, mel_b = b.decode(ids_b, il_b, return_audios=True, noise=noise)
    assert torch.equal(ids_a, ids_b) and torch.equal(il_a, il_b)
    assert torch.equal(mel_a, mel_b) and torch.equal(wav_a, wav_b)


# ------------------------------------------------------------------------------------ torch.ops.dmel_hip.* (dispatcher registration)
def test_torch_ops_are_registered_and_match_the_oracle(dev):
    """Every hot-path op is callable as torch.ops.dmel_hip.<name> (torch.library registration over the C ABI; the reference exposes its
    one native op the same way, anti_alias_activation.cpp:19-23 / load.py:31-48): one check per op against the CPU oracle, plus
    autograd through the ops that have a native backward, and a fake-tensor (shape propagation) call for each."""
    import torch.nn.functional as F
    from oracle import ref_cpu
    from conftest import rel_err
    import dmel_codec_amd.torch_ops  # noqa: F401
    ops = torch.ops.dmel_hip
    g = torch.Generator().manual_seed(0)
    # --- anti_alias_activation_forward == fwd_cuda(input, up_filter, down_filter, alpha, beta), log-scale parameters; different filters
    x = torch.randn(2, 6, 333, generator=g)
    alpha, beta = torch.randn(6, generator=g) * 0.3, torch.randn(6, generator=g) * 0.3
    up = ref_cpu.aa_filter12()
    down = up.flip(-1) * 0.9 + 0.01                              # NOT the same taps: fwd_cuda takes two filters
    ref = ref_cpu.activation1d(x, alpha, beta, up, down, logscale=True)
    y = ops.anti_alias_activation_forward(x.to(dev), up.to(dev), down.to(dev), alpha.to(dev), beta.to(dev))
    assert rel_err(y, ref) < 1e-5
    # --- aa_snake with autograd (native backward kernel) against autograd through the oracle in float64
    x64, a64, b64 = x.double().requires_grad_(), alpha.double().requires_grad_(), beta.double().requires_grad_()
    ref_cpu.activation1d(x64, a64, b64, up.double(), down.double(), logscale=True).square().sum().backward()
    xd, ad, bd = x.to(dev).requires_grad_(), alpha.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
    ops.aa_snake(xd, ad, bd, up.to(dev), down.to(dev), True).square().sum().backward()
    assert rel_err(xd.grad, x64.grad) < 2e-5 and rel_err(ad.grad, a64.grad) < 1e-4 and rel_err(bd.grad, b64.grad) < 1e-4
    # --- conv1d_dilated + backward
    w, b = torch.randn(48, 24, 7, generator=g) / (24 * 7) ** 0.5, torch.randn(48, generator=g) * 0.1
    xc = torch.randn(3, 24, 200, generator=g)
    xc64, w64, bb64 = xc.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    F.conv1d(xc64, w64, bb64, dilation=3, padding=9).square().sum().backward()
    xcd, wd, bdv = xc.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    yc = ops.conv1d_dilated(xcd, wd, bdv, 3)
    assert rel_err(yc, F.conv1d(xc64, w64, bb64, dilation=3, padding=9).detach()) < 2e-5
    yc.square().sum().backward()
    assert rel_err(xcd.grad, xc64.grad) < 2e-5 and rel_err(wd.grad, w64.grad) < 2e-5 and rel_err(bdv.grad, bb64.grad) < 2e-5
    # --- conv_transpose1d (k = 2 * stride, padding stride / 2) and conv_post: the C-ABI rows convT1d / conv_post as ops
    for stride, Ci, Co, Tt in ((8, 40, 24, 37), (2, 16, 8, 101), (4, 70, 33, 5)):
        wt, bt = torch.randn(Ci, Co, 2 * stride, generator=g) / (Ci * 2) ** 0.5, torch.randn(Co, generator=g) * 0.1
        xt = torch.randn(2, Ci, Tt, generator=g)
        ref_t = F.conv_transpose1d(xt.double(), wt.double(), bt.double(), stride=stride, padding=stride // 2)
        yt = ops.conv_transpose1d(xt.to(dev), wt.to(dev), bt.to(dev), stride)
        assert yt.shape == ref_t.shape and rel_err(yt, ref_t) < 2e-6, (stride, rel_err(yt, ref_t))
    with pytest.raises(RuntimeError, match="2 \\* stride"):
        ops.conv_transpose1d(torch.randn(1, 4, 9, device=dev), torch.randn(4, 4, 3, device=dev), None, 2)
    wp, xp = torch.randn(1, 32, 7, generator=g) * 0.05, torch.randn(3, 32, 500, generator=g)
    for act, fn in (("none", lambda v: v), ("tanh", torch.tanh), ("clamp", lambda v: v.clamp(-1, 1))):
        ref_p = fn(F.conv1d(xp.double(), wp.double(), torch.tensor([0.3], dtype=torch.float64), padding=3) * (4.0 if act == "clamp" else 1.0))
        yp = ops.conv_post(xp.to(dev) * (4.0 if act == "clamp" else 1.0), wp.to(dev), 0.3 * (4.0 if act == "clamp" else 1.0), act)
        assert yp.shape == (3, 1, 500) and rel_err(yp, ref_p) < 2e-6, act
    # --- stft_logmel
    audio = torch.randn(2, 24000, generator=g) * 0.1
    mel_ref = ref_cpu.stft_logmel(audio[:, None, :], 24000, 1024, 1024, 256, 80, 0.0, None)
    mel = ops.stft_logmel(audio.to(dev), None, 24000, 1024, 1024, 256, 80, 0.0, 0.0)
    assert mel.shape == mel_ref.shape and rel_err(mel.exp(), mel_ref.exp()) < 1e-4
    # --- module-level ops on handles owned by the mirrors (their forward() goes through the same ops)
    from dmel_codec_amd.models.modules.wavenet import WaveNet
    m = WaveNet(input_channels=10, residual_channels=32, residual_layers=3).to(dev)
    xw = torch.randn(4, 10, 50, device=dev)
    with torch.cuda.device(dev):
        h = m.native()
        ws = torch.empty(_lib_ws(h, 4, 50), dtype=torch.uint8, device=dev)
    yw = ops.wavenet_forward(h, xw, None, None, None, 1, m.output_channels, ws)
    with torch.no_grad():
        assert torch.equal(yw, m(xw))
    assert rel_err(yw, ref_cpu.wavenet_forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, "", xw.cpu(), 3)) < 1e-4
    # --- fake tensors: shapes propagate without touching the GPU library (what torch.compile's tracing needs)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        fx = torch.empty(2, 6, 333, device="cuda")
        assert ops.aa_snake(fx, torch.empty(6, device="cuda"), None, torch.empty(1, 1, 12), torch.empty(1, 1, 12), True).shape == (2, 6, 333)
        assert ops.stft_logmel(torch.empty(3, 24000, device="cuda"), None, 24000, 1024, 1024, 256, 100, 0.0, 12000.0).shape == (3, 100, 93)
        assert ops.conv1d_dilated(torch.empty(1, 24, 99, device="cuda"), torch.empty(48, 24, 7, device="cuda"), None, 3).shape == (1, 48, 99)
        assert ops.bigvgan_forward(0, torch.empty(2, 80, 10, device="cuda"), 256, torch.empty(8, dtype=torch.uint8, device="cuda")).shape == (2, 1, 2560)


def test_conv_op_with_fresh_weights_of_the_same_shape_every_call(dev):
    """the op's packed-weight cache used to be keyed on (data_ptr, _version, shape); the caching allocator re-issues the
    address of a freed weight, so a loop over fresh random weights silently reused the OLD handle from its third iteration on."""
    import torch.nn.functional as F
    from conftest import rel_err
    import dmel_codec_amd.torch_ops  # noqa: F401
    ops = torch.ops.dmel_hip
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 120, generator=g).to(dev)
    for it in range(6):
        w = (torch.randn(32, 16, 3, generator=g) / 7.0).to(dev)          # same shape, previous one freed: very likely the same address
        b = (torch.randn(32, generator=g) * 0.1).to(dev)
        y = ops.conv1d_dilated(x, w, b, 1)
        assert rel_err(y, F.conv1d(x.double(), w.double(), b.double(), padding=1)) < 2e-6, it
        del w, b
    wt_shape = (16, 8, 4)
    for it in range(4):
        wt = (torch.randn(*wt_shape, generator=g) / 5.0).to(dev)
        yt = ops.conv_transpose1d(x, wt, None, 2)
        assert rel_err(yt, F.conv_transpose1d(x.double(), wt.double(), None, stride=2, padding=1)) < 2e-6, it
        del wt
    # in-place update of a live weight (an optimiser step) is seen too
    w = (torch.randn(32, 16, 3, generator=g) / 7.0).to(dev)
    y0 = ops.conv1d_dilated(x, w, None, 1)
    w.mul_(2.0)
    assert rel_err(ops.conv1d_dilated(x, w, None, 1), 2.0 * y0) < 1e-6


def _lib_ws(h, N, T):
    from dmel_codec_amd import _lib
    return _lib.lib().dmel_wavenet_workspace_bytes(h, N, T)


def test_bf16_training_mode(dev):
    """BASELINE config 3 ("DDP bf16"): VQGAN.set_train_precision("bf16") -- convolution operands rounded to bf16, fp32 accumulate,
    everything else fp32.  Not a parity configuration (the reference's codec configs train in fp32): the check is that one training
    step stays close to the fp32 step (losses within 2 %, gradients within a few % in the L2 sense) and that ten steps reduce the mel
    loss on a fixed batch just like fp32 does."""
    from functools import partial
    opt = partial(torch.optim.AdamW, lr=2e-3, betas=(0.8, 0.99), eps=1e-5)
    sched = partial(torch.optim.lr_scheduler.LambdaLR, lr_lambda=lambda s: 1.0)
    g = torch.Generator().manual_seed(3)
    audio = (torch.randn(4, 1, 24000, generator=g) * 0.2).to(dev)
    lens = torch.tensor([24000, 20000, 24000, 12345], device=dev)
    noise = torch.randn(4, 560, 93, generator=g).to(dev)
    runs = {}
    for prec in ("fp32", "bf16"):
        codec = make_codec(4321, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3, vocoder=None, discriminator=True,
                           optimizer=opt, lr_scheduler=sched).to(dev)
        codec.set_train_precision(prec)
        gen_mel, gt, mask = codec.generator_forward(audio, lens, noise=noise)
        loss = codec.mel_loss(gen_mel, gt, mask) + ((codec.discriminator(gen_mel) - 1) ** 2).mean()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in codec.named_parameters() if p.grad is not None}
        codec.zero_grad()
        logs = [codec.training_step({"audios": audio, "audio_lengths": lens}, i, noise=noise) for i in range(10)]
        runs[prec] = (float(loss), grads, logs)
    l32, g32, logs32 = runs["fp32"]
    l16, g16, logs16 = runs["bf16"]
    assert abs(l16 - l32) < 2e-2 * abs(l32), (l16, l32)
    worst = 0.0
    for k, a in g32.items():
        b = g16[k]
        rel = float((a - b).norm() / a.norm().clamp(min=1e-20))
        worst = max(worst, rel)
        # the FSQ projections sit behind the straight-through estimator and the tanh bound: their gradients are the most sensitive
        assert rel < (0.3 if "residual_fsq" in k else 0.1), (k, rel)
    assert worst > 1e-5                     # the mode really changes the arithmetic
    m32 = [l["train/generator/loss_mel"] for l in logs32]
    m16 = [l["train/generator/loss_mel"] for l in logs16]
    assert m32[-1] < m32[0] and m16[-1] < m16[0]
    assert abs(m16[-1] - m32[-1]) < 0.1 * abs(m32[-1]), (m16, m32)


def test_two_data_parallel_ranks_of_the_real_training_step(dev):
    """The exchange step of train_codec.py on REAL native gradients with more than one rank: tools/ddp_rehearsal.py runs two ranks of
    VQGAN.training_step on this one GPU (gloo moves the CUDA tensors; RCCL refuses two ranks per device) -- native backward handing
    its flat buffer over block by block, .grad as views of it, asynchronous in-place all-reduces issued inside backward, wait before
    the clip -- and checks that after two steps all ranks hold identical parameters, equal to a single-process run whose per-rank
    gradients were averaged by hand."""
    import subprocess, sys
    from conftest import ROOT
    torch.cuda.synchronize()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_rehearsal.py")], capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-8:])
    assert r.returncode == 0 and "PASS" in r.stdout, tail
    assert '"ranks_identical": true' in r.stdout


def test_training_step_over_rccl_single_rank(dev):
    """The same rehearsal over the transport the multi-GPU run uses: backend "nccl" (= RCCL), one rank (RCCL wants one device per rank and
    the box has one).  No peer, but everything the 8-GPU job executes on this side of the wire does run: communicator creation with
    device_id, ReduceOp.AVG (gloo has none: the branch never ran before), asynchronous all-reduces issued from inside the native backward
    on RCCL's own stream, Work.wait before the clip, barrier, all_gather_object.  Parameters must equal the single-process reference."""
    import subprocess, sys
    from conftest import ROOT
    torch.cuda.synchronize()
    env = dict(os.environ, DMEL_REHEARSAL_WORLD="1", DMEL_REHEARSAL_BACKEND="nccl", DMEL_REHEARSAL_TIMEOUT="240", NCCL_SOCKET_IFNAME="lo",
               DMEL_DDP_EXERCISE_SINGLE_RANK="1")      # a group of one normally skips every collective
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_rehearsal.py")], capture_output=True, text=True, timeout=600, env=env)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0 and "PASS" in r.stdout, tail
    assert '"backend": "nccl"' in r.stdout and '"ranks_identical": true' in r.stdout, tail
    import json
    from conftest import report
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][0]
    report("[rccl] " + line)
    rec = json.loads(line)
    assert rec["n_collectives"] >= 8 and rec["arms"] == 4, line      # two steps x (discriminator pass + generator pass), block-wise messages


def test_bench_control_path_over_rccl_single_rank(dev):
    """bench.py's multi-GPU control path -- init_process_group("nccl", device_id=...), the barrier on both sides of the timed region,
    the MAX / all-gather of the per-rank times -- under the driver's launcher with one rank (DMEL_BENCH_FORCE_DIST=1 creates the process
    group although WORLD_SIZE is 1): the first run of those lines on RCCL."""
    import json, subprocess, sys
    from conftest import ROOT
    torch.cuda.synchronize()
    env = dict(os.environ, DMEL_BENCH_FORCE_DIST="1", NCCL_SOCKET_IFNAME="lo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(29500 + os.getpid() % 500), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
                        "--warmup", "1", "--cpu-budget", "0", "--median-steps", "0"], capture_output=True, text=True, timeout=600, env=env,
                       cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0, tail
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and len(line["ms_per_step_by_rank"]) == 1 and line.get("process_group") == "nccl", tail


def test_train_codec_under_torch_distributed_run_two_ranks(dev, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 train_codec.py ...` -- the launch line of INTEGRATION.md -- on this one GPU
    (DMEL_TRAIN_SHARE_DEVICE=1 points both ranks at device 0, DMEL_DIST_BACKEND=gloo because RCCL wants one GPU per rank): trainer
    process-group setup from the launcher's environment, per-rank data, gradient exchange, rank-mean of val_loss, rank-0 checkpoints
    behind a barrier."""
    import subprocess, sys
    from conftest import ROOT
    ck = str(tmp_path / "ckpt")
    over = TINY + [f"codec_ckpt_dir={ck}", "trainer.max_steps=8", "trainer.val_check_interval=2", f"tensorboard_logger.save_dir={tmp_path / 'tb'}"]
    env = dict(os.environ, DMEL_TRAIN_SHARE_DEVICE="1", DMEL_DIST_BACKEND="gloo")
    torch.cuda.synchronize()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29500 + os.getpid() % 500), os.path.join(ROOT, "train_codec.py"), *over],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-12:])
    assert r.returncode == 0, tail
    assert "training_finished" in r.stdout and r.stdout.count("start_training") == 1, tail          # rank 0 speaks once
    files = sorted(os.listdir(ck))
    assert "last.ckpt" in files, files
    ckpt = torch.load(os.path.join(ck, "last.ckpt"), map_location="cpu", weights_only=False)
    assert ckpt["global_step"] == 8 and not any("vocoder" in k for k in ckpt["state_dict"])
    lines = [l for l in open(tmp_path / "tb" / "dmel_codec_20hz" / "metrics.jsonl")]
    assert any("val_loss" in l for l in lines)
    # the first validation sample's figure and audio (codec_lit_modules.py:398-460), written by rank 0 only
    samples = tmp_path / "tb" / "dmel_codec_20hz" / "samples"
    steps = sorted(os.listdir(samples))
    assert steps and steps[0].startswith("step=")
    got = sorted(os.listdir(samples / steps[0] / "sample-0-0"))
    assert {"gt.wav", "gen.wav", "recon.wav"} <= set(got) and ("mels.png" in got or "mels.npy" in got), got


# ------------------------------------------------------------------------------------ data front end and MR-STFT loss
def test_collate_peak_normalise_and_pad_on_the_gpu(dev):
    """dataset.collate_clips_gpu (dmel_collate_peak_f32) against the host path it replaces -- peak_normalize + collate_clips, i.e.
    `librosa.util.normalize(audio) * 0.95`, longest first, right pad, (B,1,L) f32 + (1,B) i32 (dataset/lhotse_tts_dataset.py:29-32,
    :46-65): bit-identical, including a silent clip (librosa leaves it unscaled), a one-sample clip and clips longer than a tile."""
    from dmel_codec_amd.dataset import collate_clips, collate_clips_gpu, peak_normalize
    g = torch.Generator().manual_seed(5)
    lens = [24000, 9001, 1, 4097, 16000, 4096, 12345]
    clips = [torch.randn(n, generator=g) * (0.01 + i) for i, n in enumerate(lens)]
    clips[4] = torch.zeros(16000)                                  # silence: peak below `tiny` -> left as is
    clips[1][17] = -37.5                                           # the peak is a negative sample
    ref = collate_clips([peak_normalize(c) for c in clips], texts=[str(i) for i in range(len(clips))])
    out = collate_clips_gpu([c.to(dev) for c in clips], texts=[str(i) for i in range(len(clips))])
    assert out["text"] == ref["text"]
    assert out["audios"].shape == ref["audios"].shape == (7, 1, 24000) and out["audio_lengths"].shape == (1, 7)
    assert out["audio_lengths"].dtype == torch.int32 and torch.equal(out["audio_lengths"].cpu(), ref["audio_lengths"])
    assert torch.equal(out["audios"].cpu(), ref["audios"])
    assert float(out["audios"].abs().amax()) == pytest.approx(0.95, abs=1e-7)


@pytest.mark.parametrize("n_fft,hop,win,L,B", [(1024, 120, 600, 4800, 2), (2048, 240, 1200, 7200, 1), (512, 50, 240, 3000, 3),
                                               (1024, 256, 1024, 2560, 2), (512, 128, 512, 700, 1)])
def test_stft_magnitude_backward_matches_autograd(dev, n_fft, hop, win, L, B):
    """torch.ops.dmel_hip.stft_magnitude is differentiable (dmel_stft_magnitude_backward_f32: windowed DFT and its transpose as GEMMs on
    the convolution kernel + overlap-add with the reflect padding folded back): d loss / d audio against torch.autograd through the
    torch.stft restatement of the forward (oracle, utils/spectrogram.py:58-76) in float64, for a weighted sum of magnitudes and of
    log(magnitude + 0.3) -- edges (reflect fold) included.  (A bare log would hand near-empty bins a weight of 1 / |S|^2 on the fp32
    rounding of re / im: ill-conditioned in ANY fp32 implementation; the loss test below handles that case by comparison with fp32 autograd.)"""
    from oracle import ref_cpu
    from conftest import rel_err
    import dmel_codec_amd.torch_ops  # noqa: F401
    g = torch.Generator().manual_seed(n_fft + hop + L)
    y = torch.randn(B, L, generator=g) * 0.3
    T, K = L // hop, n_fft // 2 + 1
    wgt = torch.randn(B, T, K, generator=g)
    y64 = y.double().requires_grad_()
    m64 = ref_cpu.stft_magnitude(y64, n_fft, win, hop).transpose(1, 2)  # (B, K, T) -> (B, T, K), float64
    assert m64.shape == (B, T, K)
    ((m64 * wgt.double()).sum() + (m64 + 0.3).log().sum()).backward()
    yd = y.to(dev).requires_grad_()
    m = torch.ops.dmel_hip.stft_magnitude(yd, n_fft, win, hop)
    assert rel_err(m, m64.detach()) < 1e-4
    ((m * wgt.to(dev)).sum() + (m + 0.3).log().sum()).backward()
    # d|S|/d(re, im) = (re, im) / |S| is a unit vector: in near-empty bins its direction hangs on the fp32 rounding of re / im, in any
    # fp32 implementation.  The bar is therefore the float32 autograd result's own distance from float64 (as in assert_close_to_truth).
    y32 = y.clone().requires_grad_()
    m32 = ref_cpu.stft_magnitude(y32, n_fft, win, hop).transpose(1, 2)
    ((m32 * wgt).sum() + (m32 + 0.3).log().sum()).backward()
    e_gpu, e_ref = rel_err(yd.grad, y64.grad), rel_err(y32.grad, y64.grad)
    from conftest import report
    report(f"stft magnitude backward n_fft {n_fft} hop {hop}: gpu-vs-fp64 {e_gpu:.2e}, fp32-autograd-vs-fp64 {e_ref:.2e}")
    assert e_gpu < max(2e-5, 3.0 * e_ref), (e_gpu, e_ref)


def test_mrstft_loss_trains_a_waveform(dev):
    """MultiResolutionSTFTLoss is a loss: its gradient with respect to the predicted waveform matches autograd through the oracle's
    definition in float64, and a few gradient steps on the waveform itself reduce it."""
    from oracle import ref_cpu
    from conftest import rel_err
    from dmel_codec_amd.utils.mrstft import MultiResolutionSTFTLoss
    g = torch.Generator().manual_seed(21)
    target = torch.randn(2, 6000, generator=g) * 0.2
    pred = target + 0.05 * torch.randn(2, 6000, generator=g)
    p64 = pred.double().requires_grad_()
    sc64, lm64 = ref_cpu.mrstft_loss(p64, target.double())
    (sc64 + lm64).backward()
    loss = MultiResolutionSTFTLoss()
    pd = pred.to(dev).requires_grad_()
    sc, lm = loss(pd[:, None, :], target.to(dev)[:, None, :])
    assert abs(float(sc) - float(sc64)) < 1e-4 * abs(float(sc64)) and abs(float(lm) - float(lm64)) < 1e-4 * abs(float(lm64))
    (sc + lm).backward()
    # log |S| weighs near-empty bins by 1 / |S|^2: the fp32 rounding of re / im there dominates the error of any fp32 evaluation, so the
    # bar is the float32 oracle's own distance from the float64 one (the criterion of assert_close_to_truth)
    p32 = pred.clone().requires_grad_()
    sc32, lm32 = ref_cpu.mrstft_loss(p32, target)
    (sc32 + lm32).backward()
    e_gpu, e_ref = rel_err(pd.grad, p64.grad), rel_err(p32.grad, p64.grad)
    from conftest import report
    report(f"mrstft gradient: gpu-vs-fp64 {e_gpu:.2e}, fp32-autograd-vs-fp64 {e_ref:.2e}")
    assert e_gpu < max(1e-4, 3.0 * e_ref), (e_gpu, e_ref)
    w = pred.to(dev).clone().requires_grad_()
    opt = torch.optim.Adam([w], lr=2e-3)
    first = None
    for _ in range(12):
        opt.zero_grad()
        a, b = loss(w, target.to(dev))
        (a + b).backward()
        opt.step()
        first = first if first is not None else float(a + b)
    assert float(a + b) < 0.8 * first
