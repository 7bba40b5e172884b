"""Shipped hyper-parameter sets.  The reference repo contains no BigVGAN config JSON (its stage configs point at
author-local paths, config/codec/stage/pretrain.yaml:37-38); these are the public BigVGAN configurations listed in
SURVEY.md App. B, all of which instantiate with the reference's BigVGAN class."""
from __future__ import annotations

import copy

from .models.modules.bigvgan.env import AttrDict

_COMMON = {
    "resblock": "1",
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "activation": "snakebeta",
    "snake_logscale": True,
}

BIGVGAN = {
    # base 24 kHz, 14.0 M parameters (BASELINE config 2)
    "base_24k_100band": dict(_COMMON, num_mels=100, upsample_rates=[8, 8, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4],
                             upsample_initial_channel=512, use_tanh_at_final=True, use_bias_at_final=True),
    # bigvgan_v2_24khz_100band_256x, 112.4 M parameters (the reference's actual vocoder, stage/pretrain.yaml:37-38)
    "v2_24k_100band_256x": dict(_COMMON, num_mels=100, upsample_rates=[4, 4, 2, 2, 2, 2],
                                upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], upsample_initial_channel=1536,
                                use_tanh_at_final=False, use_bias_at_final=False),
    # bigvgan_v2_44khz_128band_512x, 122.2 M parameters (BASELINE config 4)
    "v2_44k_128band_512x": dict(_COMMON, num_mels=128, upsample_rates=[8, 4, 2, 2, 2, 2],
                                upsample_kernel_sizes=[16, 8, 4, 4, 4, 4], upsample_initial_channel=1536,
                                use_tanh_at_final=False, use_bias_at_final=False),
}


def bigvgan_h(name: str, **overrides) -> AttrDict:
    h = copy.deepcopy(BIGVGAN[name])
    h.update(overrides)
    return AttrDict(h)


def build_codec(sample_rate=24000, n_fft=1024, win_length=1024, hop_length=256, n_mels=100, dmel_groups=10,
                levels=(7, 5, 5), f_min=0.0, f_max=None, residual_channels=70, encoder_layers=20, decoder_layers=20,
                downsample_factor=(2, 2), vocoder: str | dict | None = "base_24k_100band", fsq_prebound=True,
                dilation_cycle=4, discriminator: bool = False, optimizer=None, lr_scheduler=None):
    """Assemble a randomly initialised VQGAN with the reference's layer shapes (config/codec/dMel_example.yaml,
    stage/pretrain.yaml): encoder WaveNet n_mels/G -> residual_channels, FSQ over G groups, decoder WaveNet
    G*residual_channels wide conditioned on the quantised latent, BigVGAN vocoder."""
    from .models.codec_lit_modules import VQGAN
    from .models.modules.bigvgan.bigvgan import BigVGAN
    from .models.modules.dowmsample_fsq import DownsampleFiniteScalarQuantize
    from .models.modules.wavenet import WaveNet
    from .utils.spectrogram import LogMelSpectrogram

    concat = residual_channels * dmel_groups
    mel = lambda: LogMelSpectrogram(sample_rate=sample_rate, n_fft=n_fft, win_length=win_length, hop_length=hop_length,
                                    n_mels=n_mels, f_min=f_min, f_max=f_max)
    voc = None
    if vocoder is not None:
        h = bigvgan_h(vocoder) if isinstance(vocoder, str) else AttrDict(copy.deepcopy(dict(vocoder)))
        h["num_mels"] = n_mels
        voc = BigVGAN(h)
    return VQGAN(
        encoder=WaveNet(input_channels=n_mels // dmel_groups, residual_channels=residual_channels,
                        residual_layers=encoder_layers, dilation_cycle=dilation_cycle),
        quantizer=DownsampleFiniteScalarQuantize(input_dim=concat, n_codebooks=1, n_groups=dmel_groups, levels=levels,
                                                 downsample_factor=downsample_factor, is_dmel=True,
                                                 fsq_prebound=fsq_prebound),
        vocoder=voc, encode_mel_transform=mel(), gt_mel_transform=mel(),
        decoder=WaveNet(input_channels=concat, output_channels=n_mels, residual_channels=concat,
                        residual_layers=decoder_layers, dilation_cycle=dilation_cycle, condition_channels=concat),
        sampling_rate=sample_rate, dmel_groups=dmel_groups, quanlity_linear=concat, dtype="float32",
        load_vocoder_ckpt=False, discriminator=_discriminator() if discriminator else None, optimizer=optimizer,
        lr_scheduler=lr_scheduler)


def _discriminator():
    from .models.modules.discriminator import Discriminator
    return Discriminator()


def oracle_cfg(codec) -> dict:
    """The plain-dict description of a VQGAN that oracle/ref_cpu.py's functions take (tests / bench only)."""
    t = codec.encode_mel_transform
    return {"sample_rate": t.sample_rate, "n_fft": t.n_fft, "win_length": t.win_length, "hop_length": t.hop_length,
            "n_mels": t.n_mels, "f_min": t.f_min, "f_max": t.spectrogram.f_max, "dmel_groups": codec.dmel_groups,
            "levels": list(codec.quantizer.levels), "downsample_factor": tuple(codec.quantizer.downsample_factor),
            "fsq_prebound": codec.quantizer.fsq_prebound, "encoder_layers": len(codec.encoder.residual_layers),
            "decoder_layers": len(codec.decoder.residual_layers) if codec.decoder is not None else 0,
            "dilation_cycle": codec.encoder.dilation_cycle}
