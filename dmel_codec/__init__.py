"""`dmel_codec` alias package: the reference's dotted module paths, served by the MI355X implementation.

A user of ishine/dmel_codec who installs this repository instead keeps their imports and Hydra `_target_` strings
(`dmel_codec.models.modules.wavenet.WaveNet`, `dmel_codec.utils.spectrogram.LogMelSpectrogram`, ...): every module
of the codec encode()/decode() path is an alias of its mirror in `dmel_codec_amd`.  Modules outside that path
(dataset, evaluation, LM, Lightning trainer glue) are not provided and raise ImportError as usual."""
import importlib
import sys

_ALIASES = {
    "dmel_codec.utils": "dmel_codec_amd.utils",
    "dmel_codec.utils.spectrogram": "dmel_codec_amd.utils.spectrogram",
    "dmel_codec.utils.utils": "dmel_codec_amd.utils.utils",
    "dmel_codec.utils.schedule": "dmel_codec_amd.utils.schedule",
    "dmel_codec.dataset": "dmel_codec_amd.dataset",
    "dmel_codec.dataset.lhotse_tts_dataset": "dmel_codec_amd.dataset.lhotse_tts_dataset",
    "dmel_codec.dataset.synthetic": "dmel_codec_amd.dataset.synthetic",
    "dmel_codec.models": "dmel_codec_amd.models",
    "dmel_codec.models.codec_lit_modules": "dmel_codec_amd.models.codec_lit_modules",
    "dmel_codec.models.lit_modules": "dmel_codec_amd.models.codec_lit_modules",   # stale path used by dMel_used.yaml:40
    "dmel_codec.models.modules": "dmel_codec_amd.models.modules",
    "dmel_codec.models.modules.wavenet": "dmel_codec_amd.models.modules.wavenet",
    "dmel_codec.models.modules.dowmsample_fsq": "dmel_codec_amd.models.modules.dowmsample_fsq",
    "dmel_codec.models.modules.firefly": "dmel_codec_amd.models.modules.firefly",
    "dmel_codec.models.modules.discriminator": "dmel_codec_amd.models.modules.discriminator",
    "dmel_codec.models.modules.bigvgan": "dmel_codec_amd.models.modules.bigvgan",
    "dmel_codec.models.modules.bigvgan.bigvgan": "dmel_codec_amd.models.modules.bigvgan.bigvgan",
    "dmel_codec.models.modules.bigvgan.activations": "dmel_codec_amd.models.modules.bigvgan.activations",
    "dmel_codec.models.modules.bigvgan.env": "dmel_codec_amd.models.modules.bigvgan.env",
    "dmel_codec.models.modules.bigvgan.utils": "dmel_codec_amd.models.modules.bigvgan.utils",
    "dmel_codec.models.modules.bigvgan.alias_free_activation": "dmel_codec_amd.models.modules.bigvgan.alias_free_activation",
    "dmel_codec.models.modules.bigvgan.alias_free_activation.torch": "dmel_codec_amd.models.modules.bigvgan.alias_free_activation",
    "dmel_codec.models.modules.bigvgan.alias_free_activation.torch.act": "dmel_codec_amd.models.modules.bigvgan.alias_free_activation.act",
    "dmel_codec.models.modules.bigvgan.alias_free_activation.cuda": "dmel_codec_amd.models.modules.bigvgan.alias_free_activation",
    "dmel_codec.models.modules.bigvgan.alias_free_activation.cuda.activation1d": "dmel_codec_amd.models.modules.bigvgan.alias_free_activation.act",
}
for _alias, _real in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_real)
