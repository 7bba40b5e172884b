"""Host logic of train_codec.py on CPU: config merge, data contract, trainer loop (optimiser-step counting, checkpoint naming, vocoder
keys stripped, resume from the newest *.ckpt), bench.py's self-launcher.  The codec itself needs the GPU (tests/test_gpu_parity.py)."""
import json
import os
import sys
import time

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_get_config_merges_defaults_and_overrides():
    from dmel_codec_amd.train_codec import _parse_overrides, get_config
    cfg = get_config(None, _parse_overrides(["trainer.max_steps=6", "model.optimizer.lr=2e-3", "model.quantizer.levels=[8, 6]"]))
    assert cfg["concat_channels_dim"] == 700 and cfg["dmel_groups"] == 10                 # from defaults[1:] (stage file)
    assert cfg["model"]["encoder"]["residual_layers"] == 20                               # stage overrides base (8)
    assert cfg["model"]["decoder"]["residual_channels"] == 700                            # ${concat_channels_dim}
    assert cfg["trainer"]["max_steps"] == 6 and cfg["model"]["optimizer"]["lr"] == 2e-3   # CLI overrides, `2e-3` read as float
    assert cfg["model"]["quantizer"]["levels"] == [8, 6]
    assert cfg["callbacks"]["model_checkpoint"]["dirpath"] == cfg["codec_ckpt_dir"]
    import dmel_codec.train_codec as alias                                                 # the reference's module path
    assert alias.get_config is get_config


def test_lightning_targets_resolve_to_the_own_trainer():
    from dmel_codec_amd import config_loader, trainer
    assert config_loader.locate("lightning.pytorch.Trainer") is trainer.Trainer
    assert config_loader.locate("lightning.pytorch.callbacks.ModelCheckpoint") is trainer.ModelCheckpoint
    ck = config_loader.instantiate({"_target_": "lightning.pytorch.callbacks.ModelCheckpoint", "dirpath": "x", "every_n_train_steps": 4,
                                    "filename": "{epoch:03d}-{step:06d}_20hz", "monitor": "val_loss", "save_last": True})
    assert ck.format_name(3, 2000) == "epoch=003-step=002000_20hz.ckpt"                   # Lightning's auto-inserted metric names


def test_batch_contract_of_the_data_side():
    """dataset/lhotse_tts_dataset.py:29-32, 46-65: peak 0.95, longest first, right-padded, (B,1,L) f32 + (1,B) i32."""
    from dmel_codec.dataset.lhotse_tts_dataset import collate_clips, peak_normalize
    from dmel_codec.dataset.synthetic import SyntheticDataModule
    g = torch.Generator().manual_seed(0)
    clips = [peak_normalize(torch.randn(n, generator=g) * s) for n, s in ((500, 3.0), (900, 0.01), (700, 1.0))]
    assert all(abs(float(c.abs().max()) - 0.95) < 1e-6 for c in clips)
    assert float(peak_normalize(torch.zeros(10)).abs().max()) == 0.0                      # silent clip: unscaled, no NaN
    b = collate_clips(clips, ["a", "b", "c"], ["p0", "p1", "p2"])
    assert b["audios"].shape == (3, 1, 900) and b["audios"].dtype == torch.float32
    assert b["audio_lengths"].tolist() == [[900, 700, 500]] and b["audio_lengths"].dtype == torch.int32
    assert b["text"] == ["b", "c", "a"] and b["audio_paths"] == ["p1", "p2", "p0"]
    assert float(b["audios"][2, 0, 500:].abs().max()) == 0.0
    dm = SyntheticDataModule(sample_rate=8000, train_max_durations=3.0, min_clip_seconds=0.5, max_clip_seconds=1.5, train_batches_per_epoch=3)
    batches = list(dm.train_dataloader())
    assert len(batches) == 3
    for bt in batches:
        lens = bt["audio_lengths"][0].tolist()
        assert lens == sorted(lens, reverse=True) and sum(lens) <= 3.0 * 8000 + 1 and bt["audios"].shape[2] == lens[0]
    again = list(SyntheticDataModule(sample_rate=8000, train_max_durations=3.0, min_clip_seconds=0.5, max_clip_seconds=1.5,
                                     train_batches_per_epoch=3).train_dataloader())
    assert all(torch.equal(a["audios"], b["audios"]) for a, b in zip(batches, again))     # seeded


class ToyCodec(torch.nn.Module):
    """The trainer-facing surface of VQGAN (two optimisers stepped by hand inside training_step, `logged`, validation_step returning
    val_loss, on_save_checkpoint dropping `vocoder.*`) on a model small enough for the CPU."""
    sampling_rate = 100
    strict_loading = False

    def __init__(self):
        super().__init__()
        self.gen = torch.nn.Linear(4, 4)
        self.disc = torch.nn.Linear(4, 1)
        self.vocoder = torch.nn.Linear(2, 2)
        self._opt = None

    def optimizers(self):
        if self._opt is None:
            self._opt = (torch.optim.AdamW(self.gen.parameters(), lr=1e-2), torch.optim.AdamW(self.disc.parameters(), lr=1e-2))
            self._sch = tuple(torch.optim.lr_scheduler.LambdaLR(o, lambda s: 1.0 / (1 + s)) for o in self._opt)
        return self._opt

    def lr_schedulers(self):
        self.optimizers()
        return self._sch

    def on_save_checkpoint(self, checkpoint):
        for k in list(checkpoint["state_dict"]):
            if "vocoder" in k:
                checkpoint["state_dict"].pop(k)

    def training_step(self, batch, batch_idx):
        og, od = self.optimizers()
        x = batch["audios"][:, 0, :4]
        ld = (self.disc(self.gen(x).detach()) ** 2).mean()
        ld.backward(); od.step(); od.zero_grad(); self._sch[1].step()
        lg = ((self.gen(x) - x) ** 2).mean()
        lg.backward(); og.step(); og.zero_grad(); self._sch[0].step()
        return {"train/generator/loss": float(lg), "train/discriminator/loss": float(ld)}

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        x = batch["audios"][:, 0, :4]
        return {"val_loss": ((self.gen(x) - x) ** 2).mean()}


def _fit(tmp_path, max_steps, seed=0):
    from dmel_codec.dataset.synthetic import SyntheticDataModule
    from dmel_codec_amd.trainer import JsonlLogger, ModelCheckpoint, ModelSummary, Trainer
    from dmel_codec_amd.utils.utils import find_lastest_ckpt
    torch.manual_seed(seed)
    model = ToyCodec()
    dm = SyntheticDataModule(sample_rate=100, train_max_durations=4.0, val_max_durations=2.0, train_batches_per_epoch=5, val_batches=2)
    ck = ModelCheckpoint(dirpath=str(tmp_path / "ckpt"), filename="{epoch:03d}-{step:06d}_20hz", monitor="val_loss", every_n_train_steps=4,
                         save_top_k=1, save_last=True)
    tr = Trainer(accelerator="cpu", max_steps=max_steps, val_check_interval=3, log_every_n_steps=1, max_epochs=10,
                 callbacks=[ck, ModelSummary()], logger=JsonlLogger(str(tmp_path / "tb"), "run"))
    tr.fit(model, dm, ckpt_path=find_lastest_ckpt(str(tmp_path / "ckpt")))
    return tr, model


def test_trainer_counts_optimizer_steps_checkpoints_and_resumes(tmp_path):
    from dmel_codec_amd.utils.utils import find_lastest_ckpt
    assert find_lastest_ckpt(None) is None and find_lastest_ckpt(str(tmp_path)) is None   # utils/utils.py:11-21
    tr, model = _fit(tmp_path, max_steps=12)
    assert tr.global_step == 12 and tr.batches_seen == 6                                  # two optimiser steps per batch
    assert tr.current_epoch == 1                                                          # 5 batches per epoch
    files = sorted(os.listdir(tmp_path / "ckpt"))
    assert "last.ckpt" in files and any(f.startswith("epoch=") and f.endswith("_20hz.ckpt") for f in files), files
    ckpt = torch.load(tmp_path / "ckpt" / "last.ckpt", weights_only=False)
    assert ckpt["global_step"] == 12 and "state_dict" in ckpt and len(ckpt["optimizer_states"]) == 2 and len(ckpt["lr_schedulers"]) == 2
    assert not any("vocoder" in k for k in ckpt["state_dict"]) and "gen.weight" in ckpt["state_dict"]      # codec_lit_modules.py:114-119
    lines = [json.loads(l) for l in open(tmp_path / "tb" / "run" / "metrics.jsonl")]
    assert any("val_loss" in l for l in lines) and any("train/generator/loss" in l for l in lines)
    # resume: newest *.ckpt by mtime, weights + optimiser + scheduler + counters restored, training continues to the new limit
    time.sleep(0.05)
    os.utime(tmp_path / "ckpt" / "last.ckpt")
    assert find_lastest_ckpt(str(tmp_path / "ckpt")).endswith("last.ckpt")
    assert ckpt["callbacks"]["ModelCheckpoint#0"]["best"] and "torch" in ckpt["rng"]     # callback state and generators travel with the file
    best_before = [tuple(b) for b in ckpt["callbacks"]["ModelCheckpoint#0"]["best"]]
    rng_probe = {}
    orig_set = torch.set_rng_state

    def spy(state):
        rng_probe["restored"] = bool(torch.equal(state, ckpt["rng"]["torch"]))
        return orig_set(state)
    torch.set_rng_state = spy
    try:
        tr2, model2 = _fit(tmp_path, max_steps=20, seed=123)                             # different init: must be overwritten by the file
    finally:
        torch.set_rng_state = orig_set
    assert rng_probe.get("restored") is True                                              # the noise stream continues where the file left it
    assert tr2.global_step == 20 and tr2.history[0]["step"] == 14
    ck2 = [c for c in tr2.callbacks if type(c).__name__ == "ModelCheckpoint"][0]
    # the resumed run knew the best score of the run it continues: at most one best file, and never a worse one than before
    assert len(ck2.best) == 1 and ck2.best[0][0] <= best_before[0][0]
    assert sum(1 for f in os.listdir(tmp_path / "ckpt") if f.startswith("epoch=")) <= 2     # the periodic file + the best one
    # the same 10 batches in one go give the same weights: resume is exact (same data order, optimiser and scheduler state)
    import shutil
    shutil.rmtree(tmp_path / "ckpt")
    tr3, model3 = _fit(tmp_path, max_steps=20)
    for (k, a), (_, b) in zip(model2.state_dict().items(), model3.state_dict().items()):
        if "vocoder" not in k:
            assert torch.allclose(a, b, atol=1e-7), k


def test_bench_self_launcher(tmp_path, capfd):
    """`python bench.py --gpus N` without torch.distributed.run: N child ranks with the launcher's environment, rank 0's stdout
    forwarded, failure of any rank -> non-zero exit, too few devices -> a clear error before anything is started."""
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, json\n"
                      "r = int(os.environ['RANK'])\n"
                      "print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')}), flush=True)\n"
                      "sys.exit(3 if (len(sys.argv) > 1 and sys.argv[1] == 'fail' and r == 1) else 0)\n")
    assert bench.spawn_ranks(2, [], script=str(script), have=2) == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert [json.loads(l) for l in out] == [{"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1"}]
    assert bench.spawn_ranks(2, ["fail"], script=str(script), have=2) == 3
    capfd.readouterr()
    assert bench.spawn_ranks(4, [], script=str(script), have=1) == 2
    assert "needs 4 devices" in capfd.readouterr().err


def test_launcher_counts_devices_without_touching_torch_cuda(monkeypatch, tmp_path):
    """bench.spawn_ranks' parent must never initialise HIP before it starts its child ranks: the device count comes from the KFD topology
    in sysfs and the *_VISIBLE_DEVICES variables, not from torch.cuda."""
    import glob as glob_mod
    import bench
    import torch

    def boom(*a, **k):
        raise AssertionError("the launcher parent called into torch.cuda")
    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    nodes = []
    for i, gfx in enumerate([0, 90500, 90500, 90500]):            # node 0 is the CPU (gfx_target_version 0)
        d = tmp_path / "nodes" / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {16 if gfx == 0 else 0}\nsimd_count 1024\ngfx_target_version {gfx}\n")
        nodes.append(str(d / "properties"))
    monkeypatch.setattr(glob_mod, "glob", lambda pattern: nodes if "kfd" in pattern else [])
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count() == 1
    assert bench.spawn_ranks(2, [], script="/nonexistent") == 2      # too few devices: refused before anything is started


def test_validation_clips_do_not_change_with_the_epoch_and_an_empty_epoch_raises(tmp_path):
    from dmel_codec.dataset.synthetic import SyntheticDataModule
    from dmel_codec_amd.trainer import Trainer
    dm = SyntheticDataModule(sample_rate=100, train_max_durations=4.0, val_max_durations=2.0, train_batches_per_epoch=2, val_batches=2)
    dm.set_epoch(0)
    v0 = [b["audios"].clone() for b in dm.val_dataloader()]
    t0 = [b["audios"].clone() for b in dm.train_dataloader()]
    dm.set_epoch(3)
    v3 = [b["audios"] for b in dm.val_dataloader()]
    t3 = [b["audios"] for b in dm.train_dataloader()]
    assert all(torch.equal(a, b) for a, b in zip(v0, v3))                  # val_loss of different epochs is measured on the same clips
    assert not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(t0, t3))
    empty = SyntheticDataModule(sample_rate=100, train_max_durations=4.0, val_max_durations=2.0, train_batches_per_epoch=0, val_batches=1)
    tr = Trainer(accelerator="cpu", max_steps=-1, max_epochs=None, callbacks=[], logger=None)
    with pytest.raises(RuntimeError, match="yielded no batch"):
        tr.fit(ToyCodec(), empty)


def test_logger_writes_the_validation_sample_the_reference_hands_to_tensorboard(tmp_path):
    """codec_lit_modules.py:398-460: mel figure + gt / gen / recon audio of the first sample -> files under the logger's directory."""
    import wave
    from dmel_codec_amd.trainer import JsonlLogger
    lg = JsonlLogger(str(tmp_path), "run")
    d = lg.log_validation_sample("sample-0-0", torch.randn(80, 50), torch.randn(80, 50),
                                 {"gt": torch.randn(12800) * 0.1, "gen": torch.randn(12800) * 0.1, "recon": 3 * torch.randn(12800)}, 24000, 12)
    assert d.endswith(os.path.join("samples", "step=000012", "sample-0-0"))
    got = set(os.listdir(d))
    assert {"gt.wav", "gen.wav", "recon.wav"} <= got and ("mels.png" in got or "mels.npy" in got)
    with wave.open(os.path.join(d, "recon.wav")) as w:
        assert w.getframerate() == 24000 and w.getnframes() == 12800 and w.getsampwidth() == 2      # clipped to [-1, 1], 16-bit PCM
