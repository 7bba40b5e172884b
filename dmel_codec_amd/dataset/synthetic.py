"""Synthetic stand-in for the reference's LhotseDataModule (dataset/lhotse_tts_dataset.py:68-218): same constructor vocabulary
(`stage`, `*_max_durations` = seconds of audio per batch per rank, `world_size`), same batch dict, no disk.  Clips are seeded band-limited
noise with random durations, peak-normalised like the reference's loader; every rank draws its own clips (the reference shards cuts
by duration across ranks with DynamicBucketingSampler(world_size=...), :184-191)."""
from __future__ import annotations

import os
from typing import Iterator, Optional

import torch

from .lhotse_tts_dataset import collate_clips, peak_normalize


class SyntheticDataModule:
    def __init__(self, stage: str = "fit", sample_rate: int = 24000, train_max_durations: float = 32.0, val_max_durations: float = 4.0,
                 min_clip_seconds: float = 1.0, max_clip_seconds: float = 1.0, train_batches_per_epoch: int = 100, val_batches: int = 2,
                 world_size: Optional[int] = None, seed: int = 1234, train_num_workers: int = 0, val_num_workers: int = 0,
                 pin_memory: bool = False, **_unused):
        assert stage in ("fit", "validate", "test"), "stage must in [fit, validate, test]"
        self.stage, self.sample_rate = stage, int(sample_rate)
        self.train_max_durations, self.val_max_durations = float(train_max_durations), float(val_max_durations)
        self.min_clip_seconds, self.max_clip_seconds = float(min_clip_seconds), float(max_clip_seconds)
        self.train_batches_per_epoch, self.val_batches = int(train_batches_per_epoch), int(val_batches)
        self.world_size = world_size
        self.seed = int(seed)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _batch(self, gen: torch.Generator, budget_seconds: float) -> dict:
        clips, total = [], 0.0
        while True:
            d = self.min_clip_seconds + (self.max_clip_seconds - self.min_clip_seconds) * float(torch.rand((), generator=gen))
            if clips and total + d > budget_seconds + 1e-9:
                break
            n = max(1, int(round(d * self.sample_rate)))
            x = torch.randn(1, 1, n, generator=gen)
            k = torch.hann_window(9, periodic=False)
            x = torch.nn.functional.conv1d(x, (k / k.sum()).view(1, 1, -1), padding=4).reshape(-1)
            clips.append(peak_normalize(x))
            total += d
            if total >= budget_seconds - 1e-9:
                break
        return collate_clips(clips)

    def _loader(self, n_batches: int, budget: float, salt: int, epoch: int) -> Iterator[dict]:
        rank = int(os.environ.get("RANK", "0"))
        for i in range(n_batches):
            gen = torch.Generator().manual_seed(self.seed + 1000003 * salt + 7919 * epoch + 104729 * rank + i)
            yield self._batch(gen, budget)

    def train_dataloader(self):
        return self._loader(self.train_batches_per_epoch, self.train_max_durations, 1, self.epoch)

    def val_dataloader(self):
        # the SAME clips at every validation: the monitored val_loss (ModelCheckpoint's top-k) compares like with like
        return self._loader(self.val_batches, self.val_max_durations, 2, 0)
