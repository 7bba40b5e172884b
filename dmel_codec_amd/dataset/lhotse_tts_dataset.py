"""Batch contract of the reference's dataset/lhotse_tts_dataset.py, without lhotse / librosa (absent here; disk I/O is outside the
hot path -- SURVEY.md section 2 #16).  What the training step depends on is kept exactly:

* every clip is peak-normalised to 0.95 (`librosa.util.normalize(audio) * 0.95`, lhotse_tts_dataset.py:29-32),
* clips of a batch are sorted by duration, longest first (:20), right-padded with zeros to the longest and stacked to
  `audios (B, 1, L) float32`; `audio_lengths` is `(1, B) int32` (:46-65)."""
from __future__ import annotations

from typing import List, Sequence

import torch


def peak_normalize(audio: torch.Tensor, peak: float = 0.95) -> torch.Tensor:
    """librosa.util.normalize (norm=inf, axis=0) * 0.95 on a mono clip: x / max|x| * 0.95, clips whose peak is below librosa's
    threshold (the dtype's `tiny`) are left unscaled (lhotse_tts_dataset.py:32)."""
    m = audio.abs().amax(dim=-1, keepdim=True)
    tiny = torch.finfo(audio.dtype).tiny
    length = torch.where(m < tiny, torch.ones_like(m), m)       # librosa: norms below the threshold become 1 (fill = None)
    return (audio / length) * peak                               # a division, then the 0.95: librosa's own order of operations


def collate_clips_gpu(clips: Sequence[torch.Tensor], texts: Sequence[str] | None = None, paths: Sequence[str] | None = None,
                      peak: float = 0.95) -> dict:
    """peak_normalize + collate_clips in two launches of one native entry point (dmel_collate_peak_f32): the decoded mono clips (1-D fp32
    CUDA tensors, wherever the decoder / resampler left them -- no concatenation) become the batch `training_step` consumes
    (lhotse_tts_dataset.py:29-32, :46-65): longest first, x / max|x| * 0.95, right-padded, `audios (B,1,L)` f32 + `audio_lengths (1,B)` i32
    on the device.  Only the sort by length runs on the host (lengths are known there; :20)."""
    from .. import _lib
    if not clips:
        raise ValueError("empty batch")
    dev = clips[0].device
    for c in clips:
        _lib.require_cuda(c, "clip")
        if c.ndim != 1 or c.dtype != torch.float32 or not c.is_contiguous() or c.device != dev:
            raise ValueError("clips must be contiguous 1-D float32 tensors on one device")
    n = [int(c.shape[0]) for c in clips]
    order = sorted(range(len(clips)), key=lambda i: -n[i])
    B, Lmax = len(clips), max(n)
    meta = torch.tensor([c.data_ptr() for c in clips] + n, dtype=torch.int64).to(dev, non_blocking=True)
    order_dev = torch.tensor(order, dtype=torch.int32).to(dev, non_blocking=True)
    audios = torch.empty(B, 1, Lmax, dtype=torch.float32, device=dev)
    lengths = torch.empty(1, B, dtype=torch.int32, device=dev)
    peaks = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().dmel_collate_peak_f32(meta.data_ptr(), meta[B:].data_ptr(), order_dev.data_ptr(), audios.data_ptr(),
                                                    lengths.data_ptr(), peaks.data_ptr(), B, Lmax, float(peak), _lib.stream_ptr()),
                   "collate_peak")
    return {"text": [texts[i] for i in order] if texts is not None else [""] * B,
            "audios": audios, "audio_lengths": lengths,
            "audio_paths": [paths[i] for i in order] if paths is not None else [""] * B}


def collate_clips(clips: Sequence[torch.Tensor], texts: Sequence[str] | None = None, paths: Sequence[str] | None = None) -> dict:
    """LhotseTTSDataset.__getitem__ + collate_fn (lhotse_tts_dataset.py:17-65) on already-loaded mono clips (1-D tensors, any device):
    sort by duration descending, right-pad, stack -> {"text", "audios" (B,1,L) f32, "audio_lengths" (1,B) i32, "audio_paths"}."""
    order = sorted(range(len(clips)), key=lambda i: -int(clips[i].shape[-1]))
    clips = [clips[i].float() for i in order]
    lens = torch.tensor([int(c.shape[-1]) for c in clips], dtype=torch.int32)
    max_length = int(lens.max())
    audios = torch.stack([torch.nn.functional.pad(c, (0, max_length - c.shape[-1])) for c in clips], dim=0)
    if audios.ndim == 2:
        audios = audios.unsqueeze(1)
    return {"text": [texts[i] for i in order] if texts is not None else [""] * len(clips),
            "audios": audios, "audio_lengths": lens.reshape(1, -1),
            "audio_paths": [paths[i] for i in order] if paths is not None else [""] * len(clips)}


class LhotseTTSDataset(torch.utils.data.Dataset):
    """lhotse_tts_dataset.py:15-65 with the audio already decoded: an item is a list of (clip, text, path)."""

    def __getitem__(self, items: List[tuple]):
        clips = [peak_normalize(torch.as_tensor(c, dtype=torch.float32)) for c, _, _ in items]
        return collate_clips(clips, [t for _, t, _ in items], [p for _, _, p in items])

    @staticmethod
    def collate_fn(batch):
        return batch[0]          # the sampler yields whole batches (lhotse_tts_dataset.py:46-65 unpacks batch[0] the same way)


class LhotseDataModule:
    """The reference's data module reads lhotse cut manifests from disk (lhotse_tts_dataset.py:68-218); lhotse and librosa are not
    part of this package's environment and disk I/O is out of the hot-path scope: configs name
    dmel_codec.dataset.synthetic.SyntheticDataModule instead, or feed `collate_clips` from their own loader."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("LhotseDataModule needs lhotse + librosa (not available here); use "
                                  "dmel_codec.dataset.synthetic.SyntheticDataModule or dataset.collate_clips with your own loader")
