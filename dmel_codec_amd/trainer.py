"""The slice of `lightning.pytorch.Trainer` that train_codec.py of the reference uses (train_codec.py:49-63), on plain torch +
torch.distributed (Lightning is not a dependency of this package): one process per GPU, RCCL gradient exchange inside
`VQGAN.training_step` (dmel_codec_amd/ddp.py), validation every `val_check_interval` batches with the `val_loss` mean over ranks
(`sync_dist=True`, codec_lit_modules.py:382-391), Lightning-layout checkpoints (`{"state_dict": ..., "optimizer_states": ...,
"lr_schedulers": ..., "global_step", "epoch"}`; `on_save_checkpoint` strips the vocoder, codec_lit_modules.py:114-119) and resume from
the newest `*.ckpt` (utils/utils.py:11-21).  `global_step` counts OPTIMISER steps as Lightning does under manual optimisation (two per
batch: discriminator + generator -- the reference's own comment, config/codec/stage/pretrain.yaml `max_steps: 10000k / 2`).

Config `_target_` strings `lightning.pytorch.Trainer`, `lightning.pytorch.callbacks.ModelCheckpoint` and
`lightning.pytorch.loggers.TensorBoardLogger` resolve to the classes below (config_loader.locate); the cosmetic callbacks
(RichProgressBar, ModelSummary) resolve to no-ops."""
from __future__ import annotations

import json
import os
import time
from typing import Any, List, Optional

import torch
import torch.distributed as dist


class Callback:
    def setup(self, trainer, model): ...
    def on_train_batch_end(self, trainer, model): ...
    def on_validation_end(self, trainer, model, metrics): ...
    def on_fit_end(self, trainer, model): ...


class RichProgressBar(Callback):
    def __init__(self, *_, **__):
        pass


class ModelSummary(Callback):
    def __init__(self, max_depth: int = 1, **_):
        self.max_depth = max_depth

    def setup(self, trainer, model):
        if trainer.is_global_zero:
            for name, child in model.named_children():
                n = sum(p.numel() for p in child.parameters())
                print(f"[summary] {name:24s} {type(child).__name__:32s} {n / 1e6:9.3f} M", flush=True)


class JsonlLogger:
    """Stand-in for lightning.pytorch.loggers.TensorBoardLogger (tensorboard is absent): one JSON line per logging step under
    save_dir/name/metrics.jsonl, same `log_metrics(metrics, step)` call."""

    def __init__(self, save_dir: str = "tb_logs", name: str = "default", log_graph: bool = False, **_):
        self.dir = os.path.join(save_dir, name)
        self._f = None

    def log_metrics(self, metrics: dict, step: int) -> None:
        if self._f is None:
            os.makedirs(self.dir, exist_ok=True)
            self._f = open(os.path.join(self.dir, "metrics.jsonl"), "a")
        self._f.write(json.dumps({"step": step, **{k: float(v) for k, v in metrics.items()}}) + "\n")
        self._f.flush()

    def log_validation_sample(self, tag: str, gt_mel, gen_mel, wavs: dict, sample_rate: int, step: int) -> str:
        """What the reference hands to TensorBoard for the first sample of the first validation batches (codec_lit_modules.py:398-460):
        a figure of the ground-truth and the re-synthesised mel and three waveforms (gt / gen / recon).  Here: files under
        save_dir/name/samples/step=<step>/<tag>/ -- mels.png (matplotlib, Agg) or mels.npy, and 16-bit PCM <key>.wav."""
        import wave
        import numpy as np
        d = os.path.join(self.dir, "samples", f"step={step:06d}", tag)
        os.makedirs(d, exist_ok=True)
        gt, gen = gt_mel.detach().float().cpu().numpy(), gen_mel.detach().float().cpu().numpy()
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, axes = plt.subplots(2, 1, figsize=(10, 6), squeeze=False)          # utils/utils.py plot_mel: one panel per mel, titled
            for ax, m, title in zip(axes[:, 0], (gt, gen), ("Ground-Truth", "Auxiliary")):
                ax.imshow(m, origin="lower", aspect="auto")
                ax.set_title(title, fontsize="medium")
                ax.tick_params(labelsize="x-small", left=False, labelleft=False)
            fig.savefig(os.path.join(d, "mels.png"), dpi=80)
            plt.close(fig)
        except Exception:
            np.save(os.path.join(d, "mels.npy"), np.stack([gt, gen]))
        for key, w in wavs.items():
            pcm = (w.detach().float().cpu().clamp(-1, 1).numpy().reshape(-1) * 32767.0).astype("<i2")
            with wave.open(os.path.join(d, f"{key}.wav"), "wb") as f:
                f.setnchannels(1)
                f.setsampwidth(2)
                f.setframerate(int(sample_rate))
                f.writeframes(pcm.tobytes())
        return d


class ModelCheckpoint(Callback):
    """lightning.pytorch.callbacks.ModelCheckpoint for the arguments the codec configs pass (dMel_example.yaml:135-144): a
    checkpoint every `every_n_train_steps` optimiser steps named by `filename` (Lightning's `{epoch:03d}-{step:06d}` ->
    `epoch=000-step=002000`), `save_last` -> last.ckpt, `save_top_k` best by `monitor` after every validation."""

    def __init__(self, dirpath: Optional[str] = None, filename: str = "{epoch}-{step}", monitor: Optional[str] = None, mode: str = "min",
                 every_n_train_steps: Optional[int] = None, save_top_k: int = 1, save_last: bool = False, verbose: bool = False, **_):
        self.dirpath, self.filename, self.monitor, self.mode = dirpath, filename, monitor, mode
        self.every_n_train_steps, self.save_top_k, self.save_last, self.verbose = every_n_train_steps, save_top_k, save_last, verbose
        self.best: List[tuple] = []       # (score, path)
        self._last_saved_step = -1

    def format_name(self, epoch: int, step: int) -> str:
        name = self.filename.replace("{epoch", "epoch={epoch").replace("{step", "step={step")
        return name.format(epoch=epoch, step=step) + ".ckpt"

    def _save(self, trainer, model, path: str) -> None:
        trainer.save_checkpoint(path)
        if self.verbose and trainer.is_global_zero:
            print(f"[ckpt] step {trainer.global_step}: saved {path}", flush=True)

    def on_train_batch_end(self, trainer, model):
        n = self.every_n_train_steps
        if not self.dirpath or not n or trainer.global_step == self._last_saved_step:
            return
        # global_step advances by two per batch: save when a multiple of n was reached or crossed by this batch
        if trainer.global_step // n > trainer.prev_global_step // n:
            self._last_saved_step = trainer.global_step
            path = os.path.join(self.dirpath, self.format_name(trainer.current_epoch, trainer.global_step))
            if self.monitor is None or self.save_top_k == -1:
                self._save(trainer, model, path)
            if self.save_last:
                self._save(trainer, model, os.path.join(self.dirpath, "last.ckpt"))

    def on_validation_end(self, trainer, model, metrics):
        if not self.dirpath or self.monitor is None or self.monitor not in metrics or self.save_top_k == 0:
            return
        score = float(metrics[self.monitor])
        key = score if self.mode == "min" else -score
        if self.save_top_k > 0 and len(self.best) >= self.save_top_k and key >= max(k for k, _ in self.best):
            return
        path = os.path.join(self.dirpath, self.format_name(trainer.current_epoch, trainer.global_step))
        self._save(trainer, model, path)
        self.best.append((key, path))
        self.best.sort()
        while self.save_top_k > 0 and len(self.best) > self.save_top_k:
            _, old = self.best.pop()
            if trainer.is_global_zero and old != path and os.path.exists(old):
                os.remove(old)

    def on_fit_end(self, trainer, model):
        if self.dirpath and self.save_last:
            self._save(trainer, model, os.path.join(self.dirpath, "last.ckpt"))

    # Lightning stores this under checkpoint["callbacks"]: without it a resumed run starts with an empty top-k list, saves its first
    # validation as "best" whatever its score and never prunes the files of the run it continues
    def state_dict(self) -> dict:
        return {"best": [list(b) for b in self.best], "last_saved_step": self._last_saved_step, "monitor": self.monitor, "mode": self.mode}

    def load_state_dict(self, state: dict) -> None:
        if state.get("monitor") == self.monitor and state.get("mode") == self.mode:
            self.best = sorted((float(k), str(p)) for k, p in state.get("best", []) if os.path.exists(str(p)))
        self._last_saved_step = int(state.get("last_saved_step", -1))


class Trainer:
    def __init__(self, accelerator: str = "gpu", devices: Any = -1, precision: Any = 32, max_steps: int = -1,
                 val_check_interval: Optional[int] = None, log_every_n_steps: int = 50, max_epochs: Optional[int] = None,
                 strategy: Optional[str] = None, callbacks: Optional[list] = None, logger: Any = None,
                 use_distributed_sampler: bool = True, limit_val_batches: Optional[int] = None, **_unused):
        self.accelerator, self.devices, self.precision = accelerator, devices, precision
        self.max_steps = -1 if max_steps is None else int(max_steps)
        self.val_check_interval = val_check_interval
        self.log_every_n_steps = max(1, int(log_every_n_steps))
        self.max_epochs = max_epochs
        self.strategy = strategy
        self.callbacks = [c for c in (callbacks or []) if isinstance(c, Callback)]
        self.logger = logger
        self.limit_val_batches = limit_val_batches
        self.global_step = 0          # optimiser steps (Lightning's definition)
        self.prev_global_step = 0
        self.batches_seen = 0
        self.batch_in_epoch = 0       # batches of the current epoch already trained on (mid-epoch resume skips them)
        self.current_epoch = 0
        self.world_size, self.global_rank, self.local_rank = 1, 0, 0
        self.device = torch.device("cpu")
        self.model = None
        self.history: List[dict] = []   # logged metrics of every batch (rank 0), for callers / tests

    @property
    def is_global_zero(self) -> bool:
        return self.global_rank == 0

    # ------------------------------------------------------------------------------------------------------ setup
    def _setup_distributed(self) -> None:
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.global_rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        want_gpu = self.accelerator in ("gpu", "cuda", "auto")
        if want_gpu:
            if not torch.cuda.is_available():
                raise RuntimeError("train_codec.py needs an MI355X: the native training path has no CPU fallback")
            # several ranks may be pointed at one device for rehearsals (DMEL_TRAIN_SHARE_DEVICE=1, gloo only: RCCL wants one GPU per rank)
            index = 0 if os.environ.get("DMEL_TRAIN_SHARE_DEVICE") else self.local_rank
            torch.cuda.set_device(index)
            self.device = torch.device("cuda", index)
        if self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("DMEL_DIST_BACKEND") or ("nccl" if want_gpu else "gloo")      # "nccl" is RCCL on ROCm
            if backend == "nccl":
                dist.init_process_group(backend, device_id=self.device)
            else:
                dist.init_process_group(backend)

    def _apply_precision(self, model) -> None:
        p = str(self.precision)
        if p in ("32", "32-true", "fp32", "float32"):
            return
        if p in ("bf16", "bf16-mixed", "bf16-true", "bfloat16"):
            if not hasattr(model, "set_train_precision"):
                raise NotImplementedError("this model has no bf16 training mode")
            model.set_train_precision("bf16")
            return
        raise NotImplementedError(f"trainer precision {self.precision!r} is not built (32 or bf16-mixed)")

    # ------------------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, path: str) -> None:
        model = self.model
        checkpoint = {
            "epoch": self.current_epoch, "global_step": self.global_step, "batches_seen": self.batches_seen,
            "batch_in_epoch": self.batch_in_epoch,
            "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
            "optimizer_states": [o.state_dict() for o in model.optimizers()],
            "lr_schedulers": [s.state_dict() for s in model.lr_schedulers()],
            "dmel_codec_amd": {"format": "lightning-layout", "version": 2},
            # callback state (top-k list) and the generators the decoder's Gaussian input and the data order are drawn from: a resumed run
            # continues the noise stream instead of restarting it
            "callbacks": {f"{type(cb).__name__}#{i}": cb.state_dict() for i, cb in enumerate(self.callbacks) if hasattr(cb, "state_dict")},
            "rng": {"torch": torch.get_rng_state(),
                    "cuda": (torch.cuda.get_rng_state(self.device) if self.device.type == "cuda" else None)},
        }
        if hasattr(model, "on_save_checkpoint"):
            model.on_save_checkpoint(checkpoint)
        if self.is_global_zero:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            tmp = path + ".tmp"
            torch.save(checkpoint, tmp)
            os.replace(tmp, path)
        if self.world_size > 1:
            dist.barrier()

    def load_checkpoint(self, path: str) -> None:
        model = self.model
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        strict = getattr(model, "strict_loading", True)
        missing, unexpected = model.load_state_dict(checkpoint["state_dict"], strict=strict)
        if unexpected:
            raise RuntimeError(f"checkpoint {path} holds keys the model does not have: {unexpected[:5]} ...")
        for o, st in zip(model.optimizers(), checkpoint.get("optimizer_states", [])):
            o.load_state_dict(st)
        for s, st in zip(model.lr_schedulers(), checkpoint.get("lr_schedulers", [])):
            s.load_state_dict(st)
        self.global_step = int(checkpoint.get("global_step", 0))
        self.prev_global_step = self.global_step
        self.current_epoch = int(checkpoint.get("epoch", 0))
        self.batches_seen = int(checkpoint.get("batches_seen", self.global_step // 2))
        self.batch_in_epoch = int(checkpoint.get("batch_in_epoch", 0))
        saved = checkpoint.get("callbacks", {})
        for i, cb in enumerate(self.callbacks):
            st = saved.get(f"{type(cb).__name__}#{i}")
            if st is not None and hasattr(cb, "load_state_dict"):
                cb.load_state_dict(st)
        rng = checkpoint.get("rng")
        if rng:
            torch.set_rng_state(rng["torch"])
            if rng.get("cuda") is not None and self.device.type == "cuda":
                torch.cuda.set_rng_state(rng["cuda"], self.device)
        if self.is_global_zero:
            print(f"[resume] {path}: epoch {self.current_epoch}, global_step {self.global_step} "
                  f"({len(missing)} keys not in the file, e.g. the vocoder's)", flush=True)

    # ------------------------------------------------------------------------------------------------------ loops
    def _to_device(self, batch: dict) -> dict:
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    def validate(self, model, datamodule) -> dict:
        was_training = model.training
        model.eval()
        if self.world_size > 1:
            from .ddp import check_parameters_in_sync
            check_parameters_in_sync(model, what=f"parameters at global step {self.global_step}")
        total, count = 0.0, 0
        for i, batch in enumerate(datamodule.val_dataloader()):
            if self.limit_val_batches is not None and i >= self.limit_val_batches:
                break
            dev_batch = self._to_device(batch)
            out = model.validation_step(dev_batch, i)
            n = batch["audios"].shape[0]
            # codec_lit_modules.py:398-460: figures and audio of the FIRST sample of the first four batches go to the logger
            if i < 4 and self.is_global_zero and hasattr(self.logger, "log_validation_sample") and "gen_aux_mels" in out:
                alen = int(torch.as_tensor(batch["audio_lengths"]).reshape(-1)[0])
                hop = getattr(getattr(model, "gt_mel_transform", None) or getattr(model, "encode_mel_transform", None), "hop_length", 256)
                ml = max(1, alen // hop)
                self.logger.log_validation_sample(
                    f"sample-{i}-0", out["gt_mels"][0, :, :ml], out["gen_aux_mels"][0, :, :ml],
                    {"gt": dev_batch["audios"][0, 0, :alen], "gen": out["gen_aux_audios"][0, 0, :alen], "recon": out["recon_audios"][0, 0, :alen]},
                    getattr(model, "sampling_rate", 24000), self.global_step)
            total += float(out["val_loss"]) * n
            count += n
        stats = torch.tensor([total, float(count)], dtype=torch.float64, device=self.device)
        if self.world_size > 1:
            dist.all_reduce(stats)                                   # sync_dist=True
        metrics = {"val_loss": float(stats[0] / stats[1].clamp(min=1))}
        if was_training:
            model.train()
        if self.logger is not None and self.is_global_zero:
            self.logger.log_metrics(metrics, self.global_step)
        for cb in self.callbacks:
            cb.on_validation_end(self, model, metrics)
        return metrics

    def fit(self, model, datamodule, ckpt_path: Optional[str] = None) -> None:
        self._setup_distributed()
        self.model = model.to(self.device)
        self._apply_precision(model)
        model.train()
        if getattr(model, "vocoder", None) is not None:
            model.vocoder.eval()
        optimizers = model.optimizers()

        def count(*_):
            self.global_step += 1
        hooks = [o.register_step_post_hook(count) for o in optimizers]
        if ckpt_path:
            self.load_checkpoint(ckpt_path)
        if self.world_size > 1:
            # what Lightning's DDP wrapper does at construction (train_codec.py:49-55): rank 0's weights are everyone's; equal seeds are
            # then a convenience.  The checksum is repeated at every validation (a skipped gradient exchange shows up there).
            from .ddp import broadcast_parameters, check_parameters_in_sync
            broadcast_parameters(model)
            check_parameters_in_sync(model)
        for cb in self.callbacks:
            cb.setup(self, model)
        t0, seen0 = time.perf_counter(), 0.0
        done = False
        try:
            while not done and (self.max_epochs is None or self.max_epochs < 0 or self.current_epoch < self.max_epochs):
                if hasattr(datamodule, "set_epoch"):
                    datamodule.set_epoch(self.current_epoch)
                resume_at = self.batch_in_epoch
                for batch_idx, batch in enumerate(datamodule.train_dataloader()):
                    if batch_idx < resume_at:
                        continue                  # already trained on before the checkpoint this run resumed from
                    if 0 <= self.max_steps <= self.global_step:
                        done = True
                        break
                    self.prev_global_step = self.global_step
                    logged = model.training_step(self._to_device(batch), batch_idx)
                    self.batches_seen += 1
                    self.batch_in_epoch = batch_idx + 1
                    seen0 += float(batch["audio_lengths"].sum()) / getattr(model, "sampling_rate", 1)
                    if self.is_global_zero:
                        self.history.append({"step": self.global_step, **(logged or {})})
                        if self.batches_seen % self.log_every_n_steps == 0 or self.log_every_n_steps == 1:
                            el = time.perf_counter() - t0
                            if self.logger is not None:
                                self.logger.log_metrics({**(logged or {}), "audio_sec_per_sec_rank0": seen0 / max(el, 1e-9)}, self.global_step)
                            print(f"[train] epoch {self.current_epoch} step {self.global_step} "
                                  + " ".join(f"{k.split('/')[-1]}={v:.5f}" for k, v in (logged or {}).items() if isinstance(v, float))
                                  + f" | {seen0 / max(el, 1e-9):.1f} audio-s/s on rank 0", flush=True)
                    for cb in self.callbacks:
                        cb.on_train_batch_end(self, model)
                    if self.val_check_interval and self.batches_seen % int(self.val_check_interval) == 0:
                        self.validate(model, datamodule)
                else:
                    if self.batch_in_epoch == 0:
                        # not one batch in a whole epoch: with max_steps = -1 and no max_epochs this loop would spin forever
                        raise RuntimeError(f"epoch {self.current_epoch}: the training dataloader yielded no batch")
                    self.current_epoch += 1
                    self.batch_in_epoch = 0
                    continue
                break
        finally:
            for h in hooks:
                h.remove()
        for cb in self.callbacks:
            cb.on_fit_end(self, model)
