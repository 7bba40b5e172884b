"""VQGAN codec: encode() / decode() on the MI355X.  Drop-in for the inference surface of
dmel_codec/models/codec_lit_modules.py (reference): same ctor kwargs, attribute names, state-dict prefixes
(`encoder.*`, `quantizer.*`, `decoder.*`, `quality_projection.*`, `vocoder.*`) and method signatures
(:462-531).  Lightning is not a dependency: this is a plain nn.Module that also carries `training_step` / `validation_step`
(:159-396) and the few trainer hooks they call (`optimizers`, `lr_schedulers`, `manual_backward`, `clip_gradients`, `log`).

Every tensor op of the path is a native HIP launch; torch only owns the memory and the stream."""
from __future__ import annotations

import ctypes as C
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
        self.set_decode_precision("fp32")

    def set_decode_precision(self, precision) -> None:
        """Arithmetic of decode()'s convolutions (decoder WaveNet + vocoder; tensors stay fp32).  "fp32" (default) is the parity path:
        fp32-grade products from the three-product fp16 split (include/dmel_hip.h, DMEL_PRECISION_FP32_F16X2 -- same error against an
        fp64 evaluation as an fp32 fma chain); "fp32_bf16x3" forces the six-product bf16 split there too; "bf16" is the opt-in
        throughput mode with bf16-rounded operands.  encode() is not affected: the ids are the interchange format and stay bit-stable
        (encoder and quantiser always run the six-product split)."""
        self._decode_precision = precision
        for m in (self.decoder, self.vocoder):
            if m is not None:
                # the decoder WaveNet handle's own "fp32" means the six-product split (an encoder is a WaveNet too): ask for the fp16 one
                m.set_precision("fp32_f16x2" if precision in ("fp32", "float32", torch.float32, 0) and m is self.decoder else precision)

    def set_train_precision(self, precision) -> None:
        """"bf16": the training paths of encoder, quantiser, decoder and discriminator run their convolutions with bf16 operands and
        fp32 accumulation (BASELINE config 3, "DDP bf16"; the trainer maps `precision: bf16-mixed` to this).  "fp32" (default) is the
        parity configuration -- the reference's codec configs train with precision 32 (config/codec/dMel_example.yaml:9)."""
        for m in (self.encoder, self.quantizer, self.decoder, self.discriminator):
            if m is not None:
                m.set_train_precision(precision)

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
            with self.grad_reducer.exchange([self.discriminator], optim_d):       # arm / backward / finish, disarmed on error
                self.manual_backward(loss_d)
        else:
            self.manual_backward(loss_d)
        if last_micro_batch:
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
            with self.grad_reducer.exchange([self.encoder, self.quantizer, self.decoder], optim_g):
                self.manual_backward(loss)                                                                       # :315
        else:
            self.manual_backward(loss)
        if last_micro_batch:
            self.clip_gradients(optim_g, gradient_clip_val=1000.0, gradient_clip_algorithm="norm")
            optim_g.step()
            optim_g.zero_grad()
            scheduler_g.step()
        return dict(self.logged)

    @torch.no_grad()
    def validation_step(self, batch, batch_idx, noise: Optional[torch.Tensor] = None):
        """codec_lit_modules.py:330-396: masked L1 between the re-synthesised and the ground-truth mel with the quality input fixed at 2
        (the condition is not masked again here, :373-379), logged as "val_loss"; then the vocoder on both mels.  The reference hands
        figures and audio of the first sample to its logger (:398-460) and returns nothing; the mirror returns the tensors and
        `Trainer.validate` hands the first sample of the first four batches to the logger (`JsonlLogger.log_validation_sample`).  `noise` (extension) injects the decoder's Gaussian input."""
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
    def streaming_decoder(self, batch: int = 1, feature_lengths: Optional[torch.Tensor] = None, return_audios: bool = True,
                          graph_chunk_tokens: Optional[int] = None, overlap_vocoder: bool = False):
        """Incremental decode with state carry (the reference decodes once, after the LM has finished,
        lm_lit_modules.py:467-471): feed token chunks as they arrive with .push(ids (B, G, n)), get audio back as soon as its right
        context exists; .finish() flushes.  See StreamingDecoder.  graph_chunk_tokens = n: once the stream has reached its steady state, a
        push of exactly n tokens is ONE HIP-graph replay instead of ~250 launches (an unbounded stream: feature_lengths must be None).
        overlap_vocoder: the vocoder of a push runs on its own stream (StreamingDecoder.wait_audio before the audio is used)."""
        return StreamingDecoder(self, batch, feature_lengths, return_audios, graph_chunk_tokens, overlap_vocoder)

    @torch.no_grad()
    def decode_stream(self, indices, feature_lengths=None, *, chunk_tokens: int = 64, noise: Optional[torch.Tensor] = None,
                      return_audios: bool = True, use_graph: bool = False, pipeline: bool = False):
        """Generator: decode() fed `chunk_tokens` tokens at a time (indices: a (B, G, T4) tensor, or any iterable of (B, G, n) chunks),
        yielding (audio | None, gen_mel) pieces whose concatenation is BIT-identical to decode() on the whole sequence.  The decoder
        WaveNet keeps the output history of every block and only ever computes new columns (dmel_wavenet_stream_step: total work 1.0x);
        the quantiser and the vocoder -- receptive fields of 3 and ~20 frames -- run on the new frames plus that context and are
        cropped.  A piece is emitted once its right context (WaveNet 75 + vocoder ~20 frames) has arrived.
        noise: (B, C, 4 T4) for reproducible runs (tensor input only), else drawn per chunk like decode() draws it.
        pipeline=True: the vocoder runs on its own stream and a piece is yielded one chunk LATER -- after the next chunk has been pulled from
        `indices` and pushed -- so that the vocoder of chunk i overlaps the decoder WaveNet of chunk i + 1 (same pieces, same bits; an
        iterator that blocks until the LM has produced the next chunk delays every piece by that long)."""
        if torch.is_tensor(indices):
            T4 = indices.shape[2]
            chunks = (indices[:, :, a:a + chunk_tokens] for a in range(0, T4, chunk_tokens))
            batch = indices.shape[0]
        else:
            chunks = iter(indices)
            first = next(chunks)
            batch = first.shape[0]
            import itertools
            chunks = itertools.chain([first], chunks)
        dec = self.streaming_decoder(batch, feature_lengths, return_audios, graph_chunk_tokens=chunk_tokens if use_graph else None,
                                     overlap_vocoder=pipeline and return_audios)
        factor = math.prod(self.quantizer.downsample_factor)
        pos = 0
        held = None                                   # pipeline: (piece, its vocoder's event), yielded after the NEXT push was enqueued

        def release(h):
            (audio, mel), ev = h
            return dec.wait_audio(audio, ev), mel

        for ids in chunks:
            n = ids.shape[2]
            nz = noise[:, :, pos * factor:(pos + n) * factor] if noise is not None else None
            pos += n
            out = dec.push(ids, noise=nz)
            if pipeline and return_audios:
                if held is not None:
                    yield release(held)
                    held = None
                if out[1].shape[-1]:
                    held = (out, dec.audio_event)
            elif out[1].shape[-1]:
                yield out
        out = dec.finish()
        if held is not None:
            yield release(held)
        if out[1].shape[-1]:
            yield (dec.wait_audio(out[0]), out[1]) if pipeline and return_audios else out

    #: mel frames of context the decode path needs on each side of a chunk for its interior to be exact:
    #: conditional WaveNet 20 layers x dilations (1,2,4,8) = 75, BigVGAN-base ~19 (conv_pre 3 + AMP/snake halos of the
    #: four stages), quantiser ConvNeXt stacks ~3.  Chunks carry a 32-token (128-frame) halo.
    STREAM_HALO_TOKENS = 32

    @torch.no_grad()
    def _decode_windows(self, indices, feature_lengths, chunk_tokens: int = 64, halo_tokens: Optional[int] = None,
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
        """decode() evaluated window by window (each window re-run with `halo_tokens` of context on both sides and cropped) and
        concatenated: same result, bounded workspace, no state.  decode_stream is the incremental form."""
        parts = list(self._decode_windows(indices, feature_lengths, chunk_tokens, halo_tokens, noise, return_audios))
        mel = torch.cat([p[1] for p in parts], dim=-1)
        if return_audios:
            return torch.cat([p[0] for p in parts], dim=-1), mel
        return mel


class StreamingDecoder:
    """State of one incremental decode (VQGAN.streaming_decoder).  Time is counted in mel frames; `origin` is the absolute frame held in
    column 0 of the state buffers (old columns are dropped once nothing reads them any more, so memory is bounded by the chunk size,
    not by the stream length)."""

    QUANT_HALO_TOKENS = 4        # ConvNeXt k7 at rates 2 and 4: 3 / 2 + 3 / 4 tokens of context on each side

    def __init__(self, codec: VQGAN, batch: int, feature_lengths, return_audios: bool, graph_chunk_tokens: Optional[int] = None,
                 overlap_vocoder: bool = False):
        if codec.decoder is None:
            raise ValueError("Decoder is not loaded")
        if overlap_vocoder and graph_chunk_tokens is not None:
            raise ValueError("overlap_vocoder and graph_chunk_tokens exclude each other: a replayed graph is one unit on its launch stream")
        if graph_chunk_tokens is not None and feature_lengths is not None:
            raise ValueError("graph_chunk_tokens needs an unbounded stream (feature_lengths=None): length masks change from push to push")
        if return_audios and codec.vocoder is None:
            raise ValueError("Vocoder is not loaded")
        self.codec, self.B, self.return_audios = codec, int(batch), return_audios
        dec = codec.decoder
        if dec.input_projection is not None:
            raise NotImplementedError("streaming needs a decoder without input projection (input_channels == residual_channels)")
        self.factor = math.prod(codec.quantizer.downsample_factor)
        self.L, self.C = len(dec.residual_layers), dec.residual_channels
        self.dils = [2 ** (i % dec.dilation_cycle) if dec.dilation_cycle else 1 for i in range(self.L)]
        self.maxdil = max(self.dils)
        self.voc_halo = codec.vocoder.receptive_field_frames() if return_audios else 0
        self.up = math.prod(codec.vocoder.h.upsample_rates) if return_audios else 1
        self.lengths = None
        if feature_lengths is not None:
            self.lengths = VQGAN._lengths(feature_lengths).to(torch.int64)
        self.tokens = None            # every token received so far is kept only as long as the quantiser's context needs it
        self.tok_origin = 0           # absolute index of tokens[..., 0]
        self.n_tok = 0
        self.origin = 0
        self.cap = 0
        self.prev = [0] * (self.L + 1)     # absolute frontier of every level
        self.z_valid = 0                    # condition / input written up to here (absolute)
        self.emitted = 0                    # mel frames handed out
        self.noise_tail = None
        self.finished = False
        # HIP-graph replay of steady-state pushes (graph_chunk_tokens): a push at batch 1 is ~250 small launches on a mostly idle chip, so
        # its time is launch count, not arithmetic.  In the steady state every push of n tokens does the same work at the same offsets
        # RELATIVE to the buffers' origin; with the buffers re-based at the start of every push those offsets are also the same
        # ADDRESSES, and the whole push -- quantiser window, every WaveNet block's new columns, vocoder window -- is one graph.
        # overlap_vocoder: the vocoder of a push runs on a stream of its own.  Within one stream the decoder WaveNet of chunk i + 1 does
        # not depend on the vocoder of chunk i (only on the WaveNet's own history), so a caller that pushes chunk i + 1 BEFORE consuming
        # the audio of chunk i (VQGAN.decode_stream(pipeline=True) does) has the two running side by side: a push costs
        # max(WaveNet, vocoder) instead of their sum.  The audio returned by push() is then valid after `wait_audio()`.
        self._overlap = bool(overlap_vocoder) and return_audios
        self._voc_stream = None
        self.audio_event = None
        self._g_n = int(graph_chunk_tokens) if graph_chunk_tokens else None
        self._rebase = self._g_n is not None      # always move the origin to the oldest column still needed (constant layout)
        self._graph = None
        self._g_sig = None                        # what the last eager push of n tokens did; two equal in a row = steady state
        self.graph_replays = 0

    # -- buffers ---------------------------------------------------------------------------------------------------
    def _ensure(self, upto: int, dev) -> None:
        """make room for absolute frames < upto: drop columns nothing will read again, grow if that is not enough"""
        need_from = max(0, min(self.prev[self.L] - self.maxdil, self.emitted - self.voc_halo))
        if self.cap and upto - self.origin <= self.cap and not (self._rebase and need_from > self.origin):
            return
        shift = need_from - self.origin
        keep = max(0, self.z_valid - need_from)
        want = upto - need_from
        if self.cap == 0 or want > self.cap:
            cap = max(512, 2 * want)
            new = dict(hist=torch.zeros(self.L + 1, self.B, self.C, cap, dtype=torch.float32, device=dev),
                       skip=torch.zeros(self.B, self.C, cap, dtype=torch.float32, device=dev),
                       cond=torch.zeros(self.B, self.codec.decoder.condition_channels, cap, dtype=torch.float32, device=dev),
                       mel=torch.zeros(self.B, self.codec.decoder.output_channels, cap, dtype=torch.float32, device=dev),
                       scratch=torch.empty(2 * self.B * self.C * cap, dtype=torch.float32, device=dev))
            if self.cap:
                for k in ("hist", "skip", "cond", "mel"):
                    new[k][..., :keep] = self.buf[k][..., shift:shift + keep]
            self.buf, self.cap = new, cap
        elif shift > 0:
            for k in ("hist", "skip", "cond", "mel"):
                self.buf[k][..., :keep] = self.buf[k][..., shift:shift + keep].clone()
        self.origin = need_from

    # -- one step ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def push(self, ids: torch.Tensor, noise: Optional[torch.Tensor] = None, final: bool = False):
        """ids (B, G, n) int (n may be 0 with final=True) -> (audio (B, 1, m * up) | None, mel (B, n_mels, m)) for the m >= 0 frames that
        became final with this chunk."""
        graphable = self._g_n is not None and not final and not self.finished and ids.shape[2] == self._g_n
        if graphable and self._graph is not None:
            return self._replay(ids, noise)
        if graphable and self._g_sig is not None and self._g_sig[0] >= 2:
            self._capture(ids.device)
            return self._replay(ids, noise)
        before = self._state()
        out = self._push_eager(ids, noise, final)
        if graphable:
            # steady state = this push moved every counter by exactly one chunk and left the carried tensors at the same sizes
            after = self._state()
            d = tuple(b - a for a, b in zip(before, after))
            nf = self._g_n * self.factor
            steady = (d == (self._g_n, self._g_n, nf, nf, nf) + (nf,) * (self.L + 1) and out[1].shape[-1] == nf)
            sig = (tuple(self.tokens.shape), tuple(self.noise_tail.shape))
            if steady and self._g_sig is not None and self._g_sig[1] == sig:
                self._g_sig = (self._g_sig[0] + 1, sig)
            else:
                self._g_sig = (1 if steady else 0, sig)
        return out

    # -- graph replay of steady-state pushes ---------------------------------------------------------------------------
    def _state(self):
        return (self.n_tok, self.tok_origin, self.origin, self.z_valid, self.emitted, *self.prev)

    def _set_state(self, st) -> None:
        self.n_tok, self.tok_origin, self.origin, self.z_valid, self.emitted = st[:5]
        self.prev = list(st[5:])

    def _capture(self, dev) -> None:
        f, n = self.factor, self._g_n
        G = self.tokens.shape[1]
        self._g_ids = torch.zeros(self.B, G, n, dtype=torch.int32, device=dev)
        self._g_noise = torch.zeros(self.B, self.C, n * f, dtype=torch.float32, device=dev)
        # carried tensors become static buffers: the graph reads them, and writes the next push's values back at its end
        self._g_tok, self._g_tail = self.tokens.contiguous().clone(), self.noise_tail.contiguous().clone()
        self.tokens, self.noise_tail = self._g_tok, self._g_tail
        st0 = self._state()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            audio, mel = self._push_eager(self._g_ids, self._g_noise, False)
            self._g_tok.copy_(self.tokens)
            self._g_tail.copy_(self.noise_tail)
        # capture records, it does not run: the counters go back to where they were; every replay then advances them by the same amounts
        self._g_delta = tuple(b - a for a, b in zip(st0, self._state()))
        self._set_state(st0)
        self.tokens, self.noise_tail = self._g_tok, self._g_tail
        self._g_out = (audio, mel)
        self._graph = g

    def _replay(self, ids: torch.Tensor, noise: Optional[torch.Tensor]):
        _lib.require_cuda(ids, "indices")
        f, n = self.factor, self._g_n
        self._g_ids.copy_(ids)
        if noise is None:
            self._g_noise.normal_()
        else:
            if noise.shape != (self.B, self.C, n * f):
                raise ValueError(f"noise must have shape {(self.B, self.C, n * f)}")
            self._g_noise.copy_(noise)
        self._graph.replay()
        self._set_state(tuple(a + d for a, d in zip(self._state(), self._g_delta)))
        self.graph_replays += 1
        audio, mel = self._g_out
        return (audio.clone() if audio is not None else None), mel.clone()

    def _push_eager(self, ids: torch.Tensor, noise: Optional[torch.Tensor] = None, final: bool = False):
        codec, f = self.codec, self.factor
        if self.finished:
            raise RuntimeError("stream already finished")
        _lib.require_cuda(ids, "indices")
        dev = ids.device
        n = ids.shape[2]
        ids = ids.to(torch.int32)
        self.tokens = ids if self.tokens is None else torch.cat([self.tokens, ids], dim=2)
        self.n_tok += n
        total_frames = self.n_tok * f
        H = self.QUANT_HALO_TOKENS
        # ---- quantiser: new condition frames [z_valid, z_new) from a token window with H tokens of context
        z_new = total_frames if final else max(self.z_valid, (self.n_tok - H) * f)
        self._ensure(total_frames, dev)
        if noise is None:
            noise = torch.randn(self.B, self.C, n * f, dtype=torch.float32, device=dev)
        elif noise.shape != (self.B, self.C, n * f):
            raise ValueError(f"noise must have shape {(self.B, self.C, n * f)}")
        noise = noise.to(dev, torch.float32)
        self.noise_tail = noise if self.noise_tail is None else torch.cat([self.noise_tail, noise], dim=2)   # frames [z_valid, total)
        if z_new > self.z_valid:
            lo_tok = max(0, self.z_valid // f - H)
            win = self.tokens[:, :, lo_tok - self.tok_origin:].contiguous()
            if self.lengths is not None:
                wl = (self.lengths.to(dev) - lo_tok).clamp(min=0, max=win.shape[2])
            else:
                wl = torch.full((self.B,), win.shape[2], dtype=torch.int64, device=dev)
            z, _ = codec.get_quantized_features_from_indices(win, wl)
            a, b = self.z_valid - lo_tok * f, z_new - lo_tok * f
            o = self.origin
            self.buf["cond"][:, :, self.z_valid - o:z_new - o] = z[:, :, a:b]
            x0 = self.noise_tail[:, :, :z_new - self.z_valid]
            if self.lengths is not None:
                t = torch.arange(self.z_valid, z_new, device=dev)[None, None, :]
                x0 = x0 * (t < (self.lengths.to(dev) * f)[:, None, None]).to(torch.float32)
            self.buf["hist"][0][:, :, self.z_valid - o:z_new - o] = x0
            self.noise_tail = self.noise_tail[:, :, z_new - self.z_valid:]
            self.z_valid = z_new
            drop = max(0, self.z_valid // f - H - self.tok_origin)
            if drop:
                self.tokens, self.tok_origin = self.tokens[:, :, drop:], self.tok_origin + drop
        # ---- decoder WaveNet: every block advances to its new frontier
        nxt = [self.z_valid]
        for d in self.dils:
            nxt.append(self.z_valid if final else max(self.prev[len(nxt)], nxt[-1] - d))
        if final or nxt[self.L] > self.prev[self.L]:
            o = self.origin
            prev = (C.c_int64 * (self.L + 1))(*[p - o for p in self.prev])
            new = (C.c_int64 * (self.L + 1))(*[p - o for p in nxt])
            dec = codec.decoder
            with torch.cuda.device(dev):
                h = dec.native()
                b = self.buf
                _lib.check(_lib.lib().dmel_wavenet_stream_step(h, b["hist"].data_ptr(), b["skip"].data_ptr(), b["cond"].data_ptr(),
                                                               b["mel"].data_ptr(), b["scratch"].data_ptr(), self.B, self.cap, prev, new,
                                                               _lib.stream_ptr()), "wavenet_stream_step")
            if self.lengths is not None and nxt[self.L] > self.prev[self.L]:
                t = torch.arange(self.prev[self.L], nxt[self.L], device=dev)[None, None, :]
                m = (t < (self.lengths.to(dev) * f)[:, None, None]).to(torch.float32)
                self.buf["mel"][:, :, self.prev[self.L] - o:nxt[self.L] - o] *= m
            self.prev = nxt
        # ---- emit: mel frames whose vocoder context exists
        ready = self.prev[self.L]
        e_new = ready if (final or not self.return_audios) else max(self.emitted, ready - self.voc_halo)
        o = self.origin
        mel = self.buf["mel"][:, :, self.emitted - o:e_new - o].clone()
        audio = None
        if self.return_audios:
            if e_new > self.emitted:
                lo = max(0, self.emitted - self.voc_halo)
                hi = min(ready, e_new + self.voc_halo)
                win = self.buf["mel"][:, :, lo - o:hi - o].contiguous()
                if self._overlap:
                    cur = torch.cuda.current_stream(dev)
                    if self._voc_stream is None:
                        self._voc_stream = torch.cuda.Stream(device=dev)
                    self._voc_stream.wait_stream(cur)               # the window copy above (and everything before it) is done
                    with torch.cuda.stream(self._voc_stream):
                        wav = codec.vocoder(win)
                        self.audio_event = torch.cuda.Event()
                        self.audio_event.record(self._voc_stream)
                    win.record_stream(self._voc_stream)
                else:
                    wav = codec.vocoder(win)
                audio = wav[:, :, (self.emitted - lo) * self.up:(e_new - lo) * self.up]
            else:
                audio = torch.empty(self.B, 1, 0, dtype=torch.float32, device=dev)
        self.emitted = e_new
        if final:
            self.finished = True
        return audio, mel

    def wait_audio(self, audio: Optional[torch.Tensor] = None, event: Optional["torch.cuda.Event"] = None):
        """overlap_vocoder: make the current stream wait for the vocoder of the last push (or of `event`, as saved from `audio_event`
        right after a push) and return `audio`, now safe to use on this stream.  No host synchronisation."""
        ev = event if event is not None else self.audio_event
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            if audio is not None and audio.numel():
                audio.record_stream(cur)
        return audio

    def finish(self):
        """no more tokens: flush everything that was waiting for right context"""
        if self.tokens is None:
            raise RuntimeError("finish() before any token")
        G = self.tokens.shape[1]
        empty = torch.empty(self.B, G, 0, dtype=torch.int32, device=self.tokens.device)
        return self.push(empty, noise=torch.empty(self.B, self.C, 0, device=self.tokens.device), final=True)
