"""Data side of train_codec.py: the batch contract of the reference's dataset/lhotse_tts_dataset.py (peak normalisation, right-pad
collate, `(1, B)` int32 lengths) and a synthetic data module with the same interface (the benchmark and the tests have no corpus)."""
from .lhotse_tts_dataset import LhotseDataModule, LhotseTTSDataset, collate_clips, collate_clips_gpu, peak_normalize  # noqa: F401
from .synthetic import SyntheticDataModule  # noqa: F401
