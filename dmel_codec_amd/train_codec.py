"""Codec training entry point.  Mirrors dmel_codec/train_codec.py (reference): `get_config()` loads config/codec/dMel_used.yaml and
merges every file under `defaults[1:]` on top (train_codec.py:12-23); `main(config)` seeds, instantiates `config.data`, `config.model`,
the callbacks that carry a `_target_`, the logger and the trainer (`use_distributed_sampler=False`), finds the newest `*.ckpt` under
`codec_ckpt_dir` and fits from it (train_codec.py:26-64).  Hydra / OmegaConf / Lightning are replaced by dmel_codec_amd.config_loader
and dmel_codec_amd.trainer (plain torch + torch.distributed, RCCL gradient exchange overlapped with backward).

    python train_codec.py                                        # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_codec.py      # one rank per GPU
    python train_codec.py --config my.yaml trainer.max_steps=6 data.train_max_durations=8 model.decoder.residual_layers=4

`key.sub=value` arguments override config values (YAML syntax for the value)."""
from __future__ import annotations

import argparse
import os
import random
import sys
from typing import Optional, Sequence

import torch
import yaml

from . import config_loader
from .utils.utils import find_lastest_ckpt

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG = os.path.join(PACKAGE_DIR, "config", "codec", "dMel_used.yaml")


def _parse_overrides(pairs: Sequence[str]) -> dict:
    out: dict = {}
    for item in pairs:
        if "=" not in item:
            raise SystemExit(f"override '{item}' is not of the form key.sub=value")
        key, value = item.split("=", 1)
        node = out
        parts = key.split(".")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = yaml.load(value, Loader=config_loader._Loader)
    return out


def get_config(path: Optional[str] = None, overrides: Optional[dict] = None) -> dict:
    """train_codec.py:12-23: base YAML + each file of `defaults[1:]` (paths relative to the base file) merged in order."""
    return config_loader.load_config(path or DEFAULT_CONFIG, overrides)


def seed_everything(seed: int) -> None:
    random.seed(seed)
    try:
        import numpy as np
        np.random.seed(seed % (2 ** 32))
    except ImportError:
        pass
    torch.manual_seed(seed)


def main(config: dict):
    rank0 = int(os.environ.get("RANK", "0")) == 0
    seed_everything(int(config.get("seed", 0)))                                               # train_codec.py:30
    datamodule = config_loader.instantiate(config["data"])                                    # :33
    model = config_loader.instantiate(config["model"], load_vocoder_ckpt=bool((config["model"].get("vocoder") or {}).get("ckpt_path")))
    callbacks = []                                                                            # :38-43
    for _, cb_conf in (config.get("callbacks") or {}).items():
        if isinstance(cb_conf, dict) and "_target_" in cb_conf:
            callbacks.append(config_loader.instantiate(cb_conf))
    logger = config_loader.instantiate(config["tensorboard_logger"]) if config.get("tensorboard_logger") else None
    trainer = config_loader.instantiate(config["trainer"], callbacks=callbacks, logger=logger, use_distributed_sampler=False)
    latest_ckpt_path = find_lastest_ckpt(config.get("codec_ckpt_dir"))                        # :57
    if rank0:
        print(f"start_training, latest_ckpt_path: {latest_ckpt_path}", flush=True)
    trainer.fit(model=model, datamodule=datamodule, ckpt_path=latest_ckpt_path)               # :59-63
    if rank0:
        print("training_finished", flush=True)
    return trainer


def cli(argv: Optional[Sequence[str]] = None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default=None, help="base YAML (default: the packaged config/codec/dMel_used.yaml)")
    ap.add_argument("overrides", nargs="*", help="key.sub=value")
    args = ap.parse_args(argv)
    return main(get_config(args.config, _parse_overrides(args.overrides)))


if __name__ == "__main__":
    cli(sys.argv[1:])
