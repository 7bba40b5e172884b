"""YAML config loading and `_target_` instantiation for the codec configs, without Hydra/OmegaConf.

Keeps the reference's schema (config/codec/dMel_used.yaml + the files named under `defaults[1:]`, merged the way
train_codec.py:12-23 merges them; `${a.b}` interpolation; `_target_` / `_partial_` nodes) so an existing YAML builds the
MI355X codec.  `_target_` strings that name the reference's classes (`dmel_codec.…`) resolve to the mirrors in
`dmel_codec_amd` (the `dmel_codec` alias package does the same for plain imports); the stale target
`dmel_codec.models.lit_modules.VQGAN` of dMel_used.yaml:40 is accepted."""
from __future__ import annotations

import copy
import functools
import importlib
import os
import re
from typing import Any, Mapping, Optional

import yaml

_INTERP = re.compile(r"\$\{([^}]+)\}")


class _Loader(yaml.SafeLoader):
    """SafeLoader that reads `1e-4` / `1e-05` as floats, as OmegaConf does (YAML 1.1 wants a dot in the mantissa;
    the reference's configs write `lr: 1e-4`)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?|\.[0-9_]+(?:[eE][-+]?[0-9]+)?|[0-9][0-9_]*[eE][-+]?[0-9]+"
               r"|[-+]?\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))
_ALIASES = {"dmel_codec.models.lit_modules": "dmel_codec_amd.models.codec_lit_modules"}
# Lightning's trainer / callbacks / logger named by the reference's configs (dMel_example.yaml:4-14,127-157): Lightning is not a
# dependency; the slice train_codec.py uses lives in dmel_codec_amd.trainer
_TARGET_ALIASES = {
    "lightning.pytorch.Trainer": "dmel_codec_amd.trainer.Trainer",
    "lightning.pytorch.callbacks.ModelCheckpoint": "dmel_codec_amd.trainer.ModelCheckpoint",
    "lightning.pytorch.callbacks.RichProgressBar": "dmel_codec_amd.trainer.RichProgressBar",
    "lightning.pytorch.callbacks.ModelSummary": "dmel_codec_amd.trainer.ModelSummary",
    "lightning.pytorch.loggers.TensorBoardLogger": "dmel_codec_amd.trainer.JsonlLogger",
}


def merge(base: dict, over: Mapping) -> dict:
    """Recursive dict merge with OmegaConf.merge semantics for mappings (lists and scalars are replaced)."""
    out = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, Mapping) and isinstance(out.get(k), dict):
            out[k] = merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def load_config(path: str, overrides: Optional[Mapping] = None) -> dict:
    """Load `path`; merge every file listed in its `defaults[1:]` (paths relative to the file's directory, as in
    train_codec.py:16-18); apply `overrides`; resolve `${...}` interpolations."""
    with open(path) as f:
        cfg = yaml.load(f, Loader=_Loader) or {}
    base_dir = os.path.dirname(os.path.abspath(path))
    for extra in (cfg.get("defaults") or [])[1:]:
        p = os.path.join(base_dir, extra)
        if not os.path.exists(p):
            raise FileNotFoundError(f"config '{path}' lists defaults entry '{extra}', which does not exist at {p}")
        with open(p) as f:
            cfg = merge(cfg, yaml.load(f, Loader=_Loader) or {})
    if overrides:
        cfg = merge(cfg, overrides)
    # directory of the shipped BigVGAN hyper-parameter JSONs (the reference ships none); usable as ${bigvgan_config_dir}
    cfg.setdefault("bigvgan_config_dir", os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "bigvgan"))
    return resolve(cfg)


def _lookup(root: Mapping, dotted: str) -> Any:
    node: Any = root
    for part in dotted.split("."):
        if not isinstance(node, Mapping) or part not in node:
            raise KeyError(f"interpolation ${{{dotted}}}: '{part}' not found")
        node = node[part]
    return node


def resolve(cfg: dict) -> dict:
    """Replace `${a.b.c}` by the value at that absolute path (whole-string references keep the value's type)."""
    def walk(node):
        if isinstance(node, dict):
            return {k: walk(v) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v) for v in node]
        if isinstance(node, str):
            m = _INTERP.fullmatch(node)
            if m:
                return walk(_lookup(cfg, m.group(1).strip()))
            return _INTERP.sub(lambda mm: str(walk(_lookup(cfg, mm.group(1).strip()))), node)
        return node
    out = walk(cfg)
    missing = [p for p, v in _flatten(out) if v == "???"]
    if missing:
        raise ValueError(f"mandatory config values still '???': {', '.join(missing)}")
    return out


def _flatten(node, prefix=""):
    if isinstance(node, dict):
        for k, v in node.items():
            if k in ("trainer", "data", "callbacks", "tensorboard_logger"):
                continue            # training control plane: not instantiated by this package
            yield from _flatten(v, f"{prefix}{k}.")
    elif isinstance(node, list):
        for i, v in enumerate(node):
            yield from _flatten(v, f"{prefix}{i}.")
    else:
        yield prefix[:-1], node


def locate(target: str):
    target = _TARGET_ALIASES.get(target, target)
    module, _, name = target.rpartition(".")
    module = _ALIASES.get(module, module)
    if module == "dmel_codec" or module.startswith("dmel_codec."):
        module = "dmel_codec_amd" + module[len("dmel_codec"):]
    try:
        return getattr(importlib.import_module(module), name)
    except (ImportError, AttributeError) as e:
        raise ImportError(f"cannot resolve _target_ '{target}' (looked for {module}.{name}): {e}") from e


def instantiate(node: Any, **extra) -> Any:
    """hydra.utils.instantiate for the subset the codec configs use: nested `_target_` nodes are built depth-first,
    `_partial_: true` returns functools.partial, other mappings/lists are returned converted."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, Mapping):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    kwargs = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_", "_convert_")}
    kwargs.update(extra)
    fn = locate(node["_target_"])
    if node.get("_partial_", False):
        return functools.partial(fn, **kwargs)
    return fn(**kwargs)


def build_codec_from_config(path: str, overrides: Optional[Mapping] = None, load_vocoder_ckpt: bool = True):
    """The `model` node of a codec YAML -> VQGAN (what train_codec.py:36 instantiates)."""
    cfg = load_config(path, overrides)
    return instantiate(cfg["model"], load_vocoder_ckpt=load_vocoder_ckpt)
