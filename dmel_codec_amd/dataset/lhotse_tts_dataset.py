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
    scale = torch.where(m > tiny, 1.0 / m.clamp(min=tiny), torch.ones_like(m))
    return audio * scale * peak


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
