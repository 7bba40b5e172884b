"""VQGAN codec: encode() / decode() on the MI355X.  Drop-in for the inference surface of
dmel_codec/models/codec_lit_modules.py (reference): same ctor kwargs, attribute names, state-dict prefixes
(`encoder.*`, `quantizer.*`, `decoder.*`, `quality_projection.*`, `vocoder.*`) and method signatures
(:462-531).  Lightning is not a dependency: this is a plain nn.Module that also carries `training_step` / `validation_step`
(:159-396) and the few trainer hooks they call (`optimizers`, `lr_schedulers`, `manual_backward`, `clip_gradients`, `log`).

Every tensor op of the path is a native HIP launch; torch only owns the memory and the stream."""
from __future__ import annotations

import math
from pathlib import Path
from typing import Callable, Optional

import torch
from torch import nn

from .. import _lib
from ..utils.spectrogram import LogMelSpectrogram
from ..utils.utils import avg_with_mask, sequence_mask
from .modules.bigvgan.bigvgan import BigVGAN
from .modules.dowmsample_fsq import DownsampleFiniteScalarQuantize
from .modules.wavenet import WaveNet


class VQGAN(nn.Module):
    def __init__(self, encoder: WaveNet, quantizer: DownsampleFiniteScalarQuantize, vocoder: Optional[BigVGAN],
                 encode_mel_transform: LogMelSpectrogram, gt_mel_transform: Optional[LogMelSpectrogram] = None,
                 optimizer: Callable | None = None, lr_scheduler: Callable | None = None, discriminator=None,
                 decoder: WaveNet | None = None, weight_adv: float = 1.0, weight_vq: float = 1.0,
                 weight_mel: float = 1.0, sampling_rate: int = 44100, freeze_encoder: bool = False,
                 dmel_groups: int = 0, quanlity_linear: int = 768, dtype: torch.dtype | str = "bfloat16",
                 accumulate_grad: int = 1, load_vocoder_ckpt: bool = True):
        super().__init__()
        # codec_lit_modules.py:52-56.  The native path computes in fp32 (the reference's codec-training dtype,
        # dMel_example.yaml:47); a bf16 request (LM configs) is honoured at the API boundary only.
        self.encode_dtype = getattr(torch, dtype) if isinstance(dtype, str) else dtype
        self.optimizer_builder, self.lr_scheduler_builder = optimizer, lr_scheduler
        self.encoder, self.quantizer = encoder, quantizer
        # codec_lit_modules.py:66-84: the reference keeps vocoder/decoder/discriminator only when the vocoder
        # checkpoint exists on disk.  load_vocoder_ckpt=False (extension) keeps randomly initialised modules, which
        # is what the parity tests and the synthetic benchmark need (no weights ship with the reference).
        if vocoder is not None and load_vocoder_ckpt and vocoder.ckpt_path and Path(vocoder.ckpt_path).exists():
            vocoder.load_state_dict(torch.load(vocoder.ckpt_path, map_location="cpu")["generator"], strict=True)
            keep = True
        else:
            keep = not load_vocoder_ckpt
        if keep:
            self.vocoder = vocoder.eval() if vocoder is not None else None
            if self.vocoder is not None:
                for p in self.vocoder.parameters():
                    p.requires_grad = False
            self.decoder, self.discriminator = decoder, discriminator
        else:
            self.vocoder, self.decoder, self.discriminator = None, None, None
        self.encode_mel_transform = encode_mel_transform
        self.gt_mel_transform = gt_mel_transform
        self.quality_projection = nn.Linear(1, quanlity_linear)   # codec_lit_modules.py:89
        self.weight_adv, self.weight_vq, self.weight_mel = weight_adv, weight_vq, weight_mel
        self.sampling_rate = sampling_rate
        self.strict_loading = False
        if freeze_encoder:
            for p in list(self.encoder.parameters()) + list(self.quantizer.parameters()):
                p.requires_grad = False
        self.dmel_groups = dmel_groups
        self.accumulate_grad = accumulate_grad
        if dmel_groups <= 0:
            raise NotImplementedError("only the dMel layout (dmel_groups > 0) works in the reference (SURVEY.md App. C)")

    def set_decode_precision(self, precision) -> None:
        """Opt-in throughput mode for decode(): "bf16" runs the decoder WaveNet and the vocoder convolutions with
        bf16-rounded operands and fp32 accumulation (tensors stay fp32); "fp32" (default) is the parity path.
        encode() is not affected: the ids are the interchange format and stay bit-stable."""
        for m in (self.decoder, self.vocoder):
            if m is not None:
                m.set_precision(precision)

    @property
    def device(self) -> torch.device:
        return self.quality_projection.weight.device

    def expand_mask(self, mask_matrix):
        return mask_matrix.repeat_interleave(self.dmel_groups, dim=0)

    @staticmethod
    def _lengths(v: torch.Tensor) -> torch.Tensor:
        return v.squeeze(0) if v.ndim == 2 else v      # collate emits (1, B): utils/utils.py:50-51

    # ------------------------------------------------------------------------------ training (generator half)
    def generator_forward(self, audios, audio_lengths, noise: Optional[torch.Tensor] = None):
        """The generator half of training_step (codec_lit_modules.py:164-211), differentiable through the native training paths of
        the encoder, the quantiser (straight-through FSQ) and the decoder: returns gen_mel, gt_mels (masked), mel_masks_float_conv.
        `noise` (extension) injects the decoder's Gaussian input, which the reference draws internally (:206)."""
        if self.decoder is None:
            raise ValueError("Decoder is not loaded")
        audios = audios.float()
        audio_lengths = self._lengths(audio_lengths)
        with torch.no_grad():                                                                  # :170-174
            encode_mels = self.encode_mel_transform(audios)
            gt_mels = (self.gt_mel_transform or self.encode_mel_transform)(audios)
            quality = ((gt_mels.mean(-1) > -8).sum(-1) - 90) / 10
            quality = quality.unsqueeze(-1)
        mel_lengths = audio_lengths.to(gt_mels.device) // (self.gt_mel_transform or self.encode_mel_transform).hop_length
        mel_masks = sequence_mask(mel_lengths, gt_mels.shape[2])                                # :176-179
        mel_masks_float_conv = mel_masks[:, None, :].to(torch.float32)
        gt_mels = gt_mels * mel_masks_float_conv
        dmel_masks = self.expand_mask(mel_masks_float_conv)                                     # :182-190
        batch_size, num_mels, time_size = encode_mels.shape
        encode_dmels = encode_mels.contiguous().view(batch_size * self.dmel_groups, num_mels // self.dmel_groups, time_size)
        encode_dmels = encode_dmels * dmel_masks
        encoded_features = self.encoder(encode_dmels) * dmel_masks
        vq_result = self.quantizer(encoded_features)                                            # :197
        vq_recon_features = vq_result.z * mel_masks_float_conv                                  # :199-202
        vq_recon_features = vq_recon_features + self.quality_projection(quality.to(torch.float32))[:, :, None]
        if noise is None:
            noise = torch.randn_like(vq_recon_features)
        gen_mel = self.decoder(noise * mel_masks_float_conv, condition=vq_recon_features * mel_masks_float_conv) * mel_masks_float_conv
        return gen_mel, gt_mels, mel_masks_float_conv

    @staticmethod
    def mel_loss(gen_mel, gt_mels, mel_masks_float_conv):
        """codec_lit_modules.py:246-263: band-weighted masked L1."""
        mel_distance = (gen_mel - gt_mels).abs()
        low = avg_with_mask(mel_distance[:, :40, :], mel_masks_float_conv)
        mid = avg_with_mask(mel_distance[:, 40:70, :], mel_masks_float_conv)
        high = avg_with_mask(mel_distance[:, 70:, :], mel_masks_float_conv)
        allb = avg_with_mask(mel_distance, mel_masks_float_conv)
        return (low * 0.6 + mid * 0.3 + high * 0.1) * 0.5 + allb * 0.5

    # Lightning's trainer owns optimizers / schedulers / logging in the reference; the mirror is a plain nn.Module, so the few hooks
    # training_step uses are provided here with the same names and the same effects.
    def configure_optimizers(self):
        """codec_lit_modules.py:121-154 (same return structure)."""
        import itertools
        optimizer_generator = self.optimizer_builder(itertools.chain(self.encoder.parameters(), self.quantizer.parameters(),
                                                                     self.decoder.parameters(), self.quality_projection.parameters()))
        optimizer_discriminator = self.optimizer_builder(self.discriminator.parameters())
        lr_scheduler_generator = self.lr_scheduler_builder(optimizer_generator)
        lr_scheduler_discriminator = self.lr_scheduler_builder(optimizer_discriminator)
        return ({"optimizer": optimizer_generator,
                 "lr_scheduler": {"scheduler": lr_scheduler_generator, "interval": "step", "name": "optimizer/generator"}},
                {"optimizer": optimizer_discriminator,
                 "lr_scheduler": {"scheduler": lr_scheduler_discriminator, "interval": "step", "name": "optimizer/discriminator"}})

    def _trainer_state(self):
        if getattr(self, "_opt_state", None) is None:
            cfg = self.configure_optimizers()
            self._opt_state = ([c["optimizer"] for c in cfg], [c["lr_scheduler"]["scheduler"] for c in cfg])
        return self._opt_state

    def optimizers(self):
        return tuple(self._trainer_state()[0])

    def lr_schedulers(self):
        return tuple(self._trainer_state()[1])

    @staticmethod
    def manual_backward(loss):
        loss.backward()

    @staticmethod
    def clip_gradients(optimizer, gradient_clip_val, gradient_clip_algorithm="norm"):
        assert gradient_clip_algorithm == "norm"
        torch.nn.utils.clip_grad_norm_([p for grp in optimizer.param_groups for p in grp["params"]], gradient_clip_val)

    @property
    def grad_reducer(self):
        """The exchange step of data-parallel training (what Lightning's DDP wrapper does during manual_backward in the reference,
        train_codec.py:49-55): dmel_codec_amd.ddp.GradReducer -- in-place RCCL all-reduce of the native flat gradient buffers, one
        bucket per decoder WaveNet block, issued from inside backward in reverse layer order and waited for before the clip.
        Inactive (gradients flow through autograd as usual) without an initialised process group or with a single rank."""
        if getattr(self, "_grad_reducer", None) is None:
            from ..ddp import GradReducer
            self._grad_reducer = GradReducer()
        return self._grad_reducer

    def on_save_checkpoint(self, checkpoint):
        """codec_lit_modules.py:114-119: the (frozen, separately distributed) vocoder is not saved with the codec."""
        state_dict = checkpoint["state_dict"]
        for name in list(state_dict.keys()):
            if "vocoder" in name:
                state_dict.pop(name)

    def log(self, name, value, **_):
        if not hasattr(self, "logged"):
            self.logged = {}
        self.logged[name] = float(value.detach()) if torch.is_tensor(value) else value

    def training_step(self, batch, batch_idx, noise: Optional[torch.Tensor] = None):
        """codec_lit_modules.py:159-327, statement by statement: discriminator step (LSGAN, masked), then generator step (band-weighted
        mel L1 + adversarial), each with manual backward / clip at 1000 / optimiser and scheduler step every `accumulate_grad` batches.
        Every network runs on its native training path (encoder, quantiser, decoder, discriminator); `noise` (extension) injects the
        decoder's Gaussian input.  Returns the dict of logged losses."""
        import torch.nn.functional as F
        if self.discriminator is None:
            raise ValueError("Discriminator is not loaded")
        optim_g, optim_d = self.optimizers()
        scheduler_g, scheduler_d = self.lr_schedulers()
        audios, audio_lengths = batch["audios"], batch["audio_lengths"]
        gen_mel, gt_mels, mel_masks_float_conv = self.generator_forward(audios, audio_lengths, noise=noise)     # :164-211
        batch_size = gen_mel.shape[0]
        loss_vq = 0.0                                                                                            # :198
        # Discriminator                                                                                          # :213-244
        real_logits = self.discriminator(gt_mels)
        fake_logits = self.discriminator(gen_mel.detach())
        d_mask = F.interpolate(mel_masks_float_conv, size=(real_logits.shape[2],), mode="nearest")
        loss_real = avg_with_mask((real_logits - 1) ** 2, d_mask)
        loss_fake = avg_with_mask(fake_logits ** 2, d_mask)
        loss_d = (loss_real + loss_fake) / self.accumulate_grad
        self.log("train/discriminator/loss", loss_d * self.accumulate_grad, batch_size=batch_size)
        # gradients are exchanged once per optimiser step, on the accumulated sum (averaging is linear: same result as the
        # reference's reduction after every backward)
        last_micro_batch = (batch_idx + 1) % self.accumulate_grad == 0
        if last_micro_batch:
            self.grad_reducer.arm([self.discriminator])
        self.manual_backward(loss_d)
        if last_micro_batch:
            self.grad_reducer.finish(optim_d)
            self.clip_gradients(optim_d, gradient_clip_val=1000.0, gradient_clip_algorithm="norm")
            optim_d.step()
            optim_d.zero_grad()
            scheduler_d.step()
        loss_mel = self.mel_loss(gen_mel, gt_mels, mel_masks_float_conv)                                         # :246-263
        fake_logits = self.discriminator(gen_mel)                                                                # :265-267
        loss_adv = avg_with_mask((fake_logits - 1) ** 2, d_mask)
        loss = (self.weight_vq * loss_vq + self.weight_mel * loss_mel + self.weight_adv * loss_adv) / self.accumulate_grad
        self.log("train/generator/loss", loss * self.accumulate_grad, batch_size=batch_size)
        self.log("train/generator/loss_vq", loss_vq, batch_size=batch_size)
        self.log("train/generator/loss_mel", loss_mel, batch_size=batch_size)
        self.log("train/generator/loss_adv", loss_adv, batch_size=batch_size)
        if last_micro_batch:
            # the discriminator is NOT armed here: like in the reference, this backward leaves gradients on its parameters that are
            # only consumed (and exchanged, as part of the sum) by the next discriminator step
            self.grad_reducer.arm([self.encoder, self.quantizer, self.decoder])
        self.manual_backward(loss)                                                                               # :315
        if last_micro_batch:
            self.grad_reducer.finish(optim_g)
            self.clip_gradients(optim_g, gradient_clip_val=1000.0, gradient_clip_algorithm="norm")
            optim_g.step()
            optim_g.zero_grad()
            scheduler_g.step()
        return dict(self.logged)

    @torch.no_grad()
    def validation_step(self, batch, batch_idx, noise: Optional[torch.Tensor] = None):
        """codec_lit_modules.py:330-396: masked L1 between the re-synthesised and the ground-truth mel with the quality input fixed at 2
        (the condition is not masked again here, :373-379), logged as "val_loss"; then the vocoder on both mels.  The reference hands
        figures and audio of the first sample to its logger (:398-460) and returns nothing; the mirror has no logger, so it returns the
        tensors a caller would log.  `noise` (extension) injects the decoder's Gaussian input."""
        if self.decoder is None:
            raise ValueError("Decoder is not loaded")
        audios, audio_lengths = batch["audios"], batch["audio_lengths"]
        audios = audios.float()
        audio_lengths = self._lengths(audio_lengths)
        gt_transform = self.gt_mel_transform or self.encode_mel_transform
        encode_mels = self.encode_mel_transform(audios)
        gt_mels = gt_transform(audios)
        mel_lengths = audio_lengths.to(gt_mels.device) // gt_transform.hop_length
        mel_masks = sequence_mask(mel_lengths, gt_mels.shape[2])
        mel_masks_float_conv = mel_masks[:, None, :].to(torch.float32)
        gt_mels = gt_mels * mel_masks_float_conv
        dmel_masks = self.expand_mask(mel_masks_float_conv)
        batch_size, num_mels, time_size = encode_mels.shape
        encode_dmels = encode_mels.contiguous().view(batch_size * self.dmel_groups, num_mels // self.dmel_groups, time_size)
        encode_dmels = encode_dmels * dmel_masks
        encoded_features = self.encoder(encode_dmels) * dmel_masks
        vq_recon_features = self.quantizer(encoded_features).z * mel_masks_float_conv
        two = torch.ones(vq_recon_features.shape[0], 1, device=vq_recon_features.device) * 2
        vq_recon_features = vq_recon_features + self.quality_projection(two)[:, :, None]
        if noise is None:
            noise = torch.randn_like(vq_recon_features)
        gen_aux_mels = self.decoder(noise * mel_masks_float_conv, condition=vq_recon_features) * mel_masks_float_conv
        loss_mel = avg_with_mask((gen_aux_mels - gt_mels).abs(), mel_masks_float_conv)
        self.log("val_loss", loss_mel, batch_size=batch_size)
        if self.vocoder is None:
            raise ValueError("Vocoder is not loaded")
        recon_audios = self.vocoder(gt_mels)
        gen_aux_audios = self.vocoder(gen_aux_mels)
        return {"val_loss": loss_mel, "gt_mels": gt_mels, "gen_aux_mels": gen_aux_mels, "recon_audios": recon_audios,
                "gen_aux_audios": gen_aux_audios}

    # ------------------------------------------------------------------------------ encode side
    @torch.no_grad()
    def encode_unquantized(self, audios, audio_lengths):
        """codec_lit_modules.py:486-513 -> features (B*G, C, T), mel_lengths (B,)"""
        audios = audios.float()
        audio_lengths = self._lengths(audio_lengths)
        hop = self.encode_mel_transform.hop_length
        mel_lengths = audio_lengths // hop
        # mels * mask fused into the STFT kernel's store; "(B, n_mels, T) -> (B*G, n_mels/G, T)" is a view
        mels = self.encode_mel_transform(audios, lengths=audio_lengths)
        B, n_mels, T = mels.shape
        x = mels.view(B * self.dmel_groups, n_mels // self.dmel_groups, T)
        ml = mel_lengths.to(mels.device)
        feats = self.encoder(x, out_lengths=ml, group_repeat=self.dmel_groups)
        return feats.to(self.encode_dtype), mel_lengths

    @torch.no_grad()
    def get_indices_from_unquantized_features(self, unquantized_features, mel_lengths):
        """codec_lit_modules.py:529-531"""
        indices_lengths = mel_lengths // math.prod(self.quantizer.downsample_factor)
        return self.quantizer.encode(unquantized_features), indices_lengths

    @torch.no_grad()
    def encode(self, audios, audio_lengths):
        """codec_lit_modules.py:462-466 -> indices (B, G, T4) int32, indices_lengths (B,)"""
        feats, mel_lengths = self.encode_unquantized(audios, audio_lengths)
        return self.get_indices_from_unquantized_features(feats, mel_lengths)

    # ------------------------------------------------------------------------------ decode side
    @torch.no_grad()
    def get_quantized_features_from_indices(self, indices, feature_lengths):
        """codec_lit_modules.py:515-527 -> z (B, G*C, 4*T4), mask (B, 1, 4*T4)"""
        feature_lengths = self._lengths(feature_lengths)
        factor = math.prod(self.quantizer.downsample_factor)
        _lib.require_cuda(indices, "indices")
        z = self.quantizer.decode(indices)
        B, Cc, T = z.shape
        lens = (feature_lengths.to(device=z.device, dtype=torch.int64) * factor).contiguous()
        w = self.quality_projection.weight.detach().reshape(-1).to(z.device, torch.float32).contiguous()
        b = self.quality_projection.bias.detach().to(z.device, torch.float32).contiguous()
        with torch.cuda.device(z.device):
            _lib.check(_lib.lib().dmel_mask_add_quality_f32(z.data_ptr(), lens.data_ptr(), w.data_ptr(), b.data_ptr(), 2.0,
                                                            B, Cc, T, _lib.stream_ptr()), "mask_add_quality")
        mask = sequence_mask(lens, T)[:, None, :].to(self.encode_dtype)
        return z, mask

    @torch.no_grad()
    def decode(self, indices, feature_lengths, return_audios=False, noise: Optional[torch.Tensor] = None):
        """codec_lit_modules.py:468-484.  noise (extension): the Gaussian decoder input the reference draws with
        torch.randn_like (:473); pass it for reproducible / parity runs."""
        if self.decoder is None:
            raise ValueError("Decoder is not loaded")
        feature_lengths = self._lengths(feature_lengths)
        factor = math.prod(self.quantizer.downsample_factor)
        z, _ = self.get_quantized_features_from_indices(indices, feature_lengths)
        if noise is None:
            noise = torch.randn_like(z)
        elif noise.shape != z.shape:
            raise ValueError(f"noise must have shape {tuple(z.shape)}")
        lens = (feature_lengths.to(device=z.device, dtype=torch.int64) * factor).contiguous()
        gen_mel = self.decoder(noise.to(z.device), condition=z, in_lengths=lens, out_lengths=lens)
        if return_audios:
            if self.vocoder is None:
                raise ValueError("Vocoder is not loaded")
            return self.vocoder(gen_mel), gen_mel
        return gen_mel

    # ------------------------------------------------------------------------------ streaming decode (extension)
    #: mel frames of context the decode path needs on each side of a chunk for its interior to be exact:
    #: conditional WaveNet 20 layers x dilations (1,2,4,8) = 75, BigVGAN-base ~19 (conv_pre 3 + AMP/snake halos of the
    #: four stages), quantiser ConvNeXt stacks ~3.  Chunks carry a 32-token (128-frame) halo.
    STREAM_HALO_TOKENS = 32

    @torch.no_grad()
    def decode_stream(self, indices, feature_lengths, chunk_tokens: int = 64, halo_tokens: Optional[int] = None,
                      noise: Optional[torch.Tensor] = None, return_audios: bool = True):
        """Generator over time chunks of decode(): yields (audio (B,1,n*256*4) | None, gen_mel (B,n_mels,n*4)) for
        successive windows of `chunk_tokens` token frames, each decoded with `halo_tokens` of context on both sides
        and cropped.  Every layer of the decode path is a finite-support convolution, so with the halo at least the
        receptive field the concatenation is BIT-identical to decode() on the whole sequence (tests): bounded memory
        for long audio, and audio can be emitted while an LM is still producing tokens (the reference decodes once
        at the end, lm_lit_modules.py:467-471).  `noise`: (B, C, 4*T4) for reproducible runs, else drawn per chunk."""
        if self.decoder is None:
            raise ValueError("Decoder is not loaded")
        halo = self.STREAM_HALO_TOKENS if halo_tokens is None else int(halo_tokens)
        feature_lengths = self._lengths(feature_lengths)
        B, G, T4 = indices.shape
        factor = math.prod(self.quantizer.downsample_factor)
        hop = self.encode_mel_transform.hop_length
        up = 1
        if return_audios:
            if self.vocoder is None:
                raise ValueError("Vocoder is not loaded")
            for u in self.vocoder.h.upsample_rates:
                up *= u
        for start in range(0, T4, chunk_tokens):
            stop = min(start + chunk_tokens, T4)
            lo, hi = max(0, start - halo), min(T4, stop + halo)
            ids = indices[:, :, lo:hi].contiguous()
            lens = (feature_lengths.to(indices.device) - lo).clamp(min=0, max=hi - lo)
            nz = noise[:, :, lo * factor:hi * factor].contiguous() if noise is not None else None
            out = self.decode(ids, lens, return_audios=return_audios, noise=nz)
            audio, mel = out if return_audios else (None, out)
            a, b = (start - lo) * factor, (stop - lo) * factor
            yield (audio[:, :, a * up:b * up] if audio is not None else None), mel[:, :, a:b]
        del hop

    @torch.no_grad()
    def decode_chunked(self, indices, feature_lengths, chunk_tokens: int = 64, halo_tokens: Optional[int] = None,
                       noise: Optional[torch.Tensor] = None, return_audios: bool = True):
        """decode() evaluated chunk by chunk (decode_stream) and concatenated: same result, bounded workspace."""
        parts = list(self.decode_stream(indices, feature_lengths, chunk_tokens, halo_tokens, noise, return_audios))
        mel = torch.cat([p[1] for p in parts], dim=-1)
        if return_audios:
            return torch.cat([p[0] for p in parts], dim=-1), mel
        return mel
