"""BigVGAN generator on the MI355X.  Drop-in for dmel_codec/models/modules/bigvgan/bigvgan.py (reference): same
class names, ctor arguments, `h` schema and state-dict keys (old-style weight norm `weight_g`/`weight_v`, or plain
`weight` after remove_weight_norm()).  forward() is one native call (csrc/modules.hip: dmel_bigvgan_forward) that
issues the fused anti-alias activations and the MFMA implicit-GEMM convolutions.
The HF-hub mixin of the reference (bigvgan.py:410-499) is network I/O and is not provided."""
from __future__ import annotations

import ctypes as C
import json

import torch
from torch import nn

from .... import _lib, torch_ops  # noqa: F401  (registers torch.ops.dmel_hip.*)
from .._native import NativeModule
from . import activations
from .alias_free_activation.act import Activation1d
from .env import AttrDict
from .utils import get_padding


def load_hparams_from_json(path) -> AttrDict:
    with open(path) as f:
        return AttrDict(json.loads(f.read()))


class _WNConv(nn.Module):
    """Parameters of a weight-normed Conv1d / ConvTranspose1d in the reference's old-style layout
    (torch.nn.utils.weight_norm: `weight_g` (n0,1,1), `weight_v`), initialised like the reference
    (normal(0, std) on the weight, then g = ||v||).  bias like nn.Conv1d's default."""

    def __init__(self, weight_shape, std: float | None = 0.01):
        super().__init__()
        if std is not None:
            v = torch.randn(weight_shape) * std            # init_weights, utils.py:45-48
        else:                                              # untouched nn.Conv1d default (conv_pre)
            v = torch.empty(weight_shape)
            nn.init.kaiming_uniform_(v, a=5 ** 0.5)
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = nn.Parameter(v)
        self.register_parameter("bias", None)

    def fold(self):
        """remove_weight_norm: replace (weight_g, weight_v) by the effective `weight`."""
        if hasattr(self, "weight_g"):
            w = torch._weight_norm(self.weight_v.detach(), self.weight_g.detach(), 0)
            del self.weight_g
            del self.weight_v
            self.weight = nn.Parameter(w)
        else:
            raise ValueError("weight norm already removed")


def _wn_conv(cin, cout, k, bias=True, std=0.01, transposed=False):
    shape = (cin, cout, k) if transposed else (cout, cin, k)
    m = _WNConv(shape, std=std)
    if bias:
        fan_in = (cout if transposed else cin) * k
        bound = 1.0 / (fan_in ** 0.5)
        m.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
    return m


class AMPBlock1(nn.Module):
    """bigvgan.py:31-147 (parameter container; forward runs inside dmel_bigvgan_forward)."""

    def __init__(self, h: AttrDict, channels: int, kernel_size: int = 3, dilation: tuple = (1, 3, 5), activation: str = None):
        super().__init__()
        self.h = h
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([_wn_conv(channels, channels, kernel_size) for _ in dilation])
        self.convs2 = nn.ModuleList([_wn_conv(channels, channels, kernel_size) for _ in dilation])
        self.num_layers = len(self.convs1) + len(self.convs2)
        if activation == "snake":
            make = lambda: activations.Snake(channels, alpha_logscale=h.snake_logscale)
        elif activation == "snakebeta":
            make = lambda: activations.SnakeBeta(channels, alpha_logscale=h.snake_logscale)
        else:
            raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")
        self.activations = nn.ModuleList([Activation1d(activation=make()) for _ in range(self.num_layers)])
        assert get_padding(kernel_size, 1) == (kernel_size - 1) // 2

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            l.fold()


class AMPBlock2(nn.Module):
    """bigvgan.py:150-241 (parameter container; forward runs inside dmel_bigvgan_forward): per dilation act -> conv -> + x."""

    def __init__(self, h: AttrDict, channels: int, kernel_size: int = 3, dilation: tuple = (1, 3, 5), activation: str = None):
        super().__init__()
        self.h = h
        self.kernel_size, self.dilation = kernel_size, tuple(dilation)
        self.convs = nn.ModuleList([_wn_conv(channels, channels, kernel_size) for _ in dilation])
        self.num_layers = len(self.convs)
        if activation == "snake":
            make = lambda: activations.Snake(channels, alpha_logscale=h.snake_logscale)
        elif activation == "snakebeta":
            make = lambda: activations.SnakeBeta(channels, alpha_logscale=h.snake_logscale)
        else:
            raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")
        self.activations = nn.ModuleList([Activation1d(activation=make()) for _ in range(self.num_layers)])

    def remove_weight_norm(self):
        for l in self.convs:
            l.fold()


class BigVGAN(NativeModule):
    """bigvgan.py:244-407."""

    _destroy_symbol = "dmel_bigvgan_destroy"
    _set_symbol = "dmel_bigvgan_set_tensor"
    _finalize_symbol = "dmel_bigvgan_finalize"
    _precision_symbol = "dmel_bigvgan_set_precision"

    def __init__(self, h: AttrDict = None, use_cuda_kernel: bool = False, h_path=None, ckpt_path=None):
        super().__init__()
        assert (h is not None) or (h_path is not None)
        if h is None:
            h = load_hparams_from_json(h_path)
        if not isinstance(h, AttrDict):
            h = AttrDict(dict(h))
        self.ckpt_path = ckpt_path
        self.h = h
        # the reference's flag selects its CUDA kernel (bigvgan.py:279-286); here the fused HIP kernel is the only path
        self.h["use_cuda_kernel"] = use_cuda_kernel
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        if h.resblock == "1":                                   # bigvgan.py:303-311
            resblock_class = AMPBlock1
        elif h.resblock == "2":
            resblock_class = AMPBlock2
        else:
            raise ValueError(f"Incorrect resblock class specified in hyperparameters. Got {h.resblock}")
        c0 = h.upsample_initial_channel
        self.conv_pre = _wn_conv(h.num_mels, c0, 7, std=None)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(nn.ModuleList([_wn_conv(c0 // (2 ** i), c0 // (2 ** (i + 1)), k, transposed=True)]))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(resblock_class(h, ch, k, d, activation=h.activation))
        if h.activation == "snake":
            post = activations.Snake(ch, alpha_logscale=h.snake_logscale)
        elif h.activation == "snakebeta":
            post = activations.SnakeBeta(ch, alpha_logscale=h.snake_logscale)
        else:
            raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")
        self.activation_post = Activation1d(activation=post)
        self.use_bias_at_final = h.get("use_bias_at_final", True)
        self.conv_post = _wn_conv(ch, 1, 7, bias=self.use_bias_at_final)
        self.use_tanh_at_final = h.get("use_tanh_at_final", True)

    def remove_weight_norm(self):
        try:
            for l in self.ups:
                for l_i in l:
                    l_i.fold()
            for l in self.resblocks:
                l.remove_weight_norm()
            self.conv_pre.fold()
            self.conv_post.fold()
        except ValueError:
            print("[INFO] Model already removed weight norm. Skipping!")

    def _create_native(self) -> int:
        h = self.h
        cfg = _lib.BigVGANConfig()
        cfg.num_mels = h.num_mels
        cfg.upsample_initial_channel = h.upsample_initial_channel
        cfg.num_upsamples = self.num_upsamples
        if self.num_upsamples > 8 or self.num_kernels > 8:
            raise NotImplementedError("more than 8 upsample stages / resblock kernels")
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.num_kernels = self.num_kernels
        for j, (k, d) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
            if len(d) != 3:
                raise NotImplementedError("AMP blocks with other than 3 dilations")
            cfg.resblock_kernel_sizes[j] = k
            for l in range(3):
                cfg.resblock_dilations[j][l] = d[l]
        cfg.snake_logscale = int(bool(h.snake_logscale))
        cfg.activation_snake = int(h.activation == "snake")
        cfg.use_tanh_at_final = int(bool(self.use_tanh_at_final))
        cfg.use_bias_at_final = int(bool(self.use_bias_at_final))
        cfg.resblock_type = 2 if h.resblock == "2" else 1
        handle = C.c_void_p()
        _lib.check(_lib.lib().dmel_bigvgan_create(C.byref(handle), C.byref(cfg)), "bigvgan_create")
        return handle.value

    def receptive_field_frames(self) -> int:
        """Mel frames of context on EACH side beyond which an output sample does not depend on the input (every layer has finite
        support): conv_pre 3 frames; per stage one input frame for the transposed conv (k = 2u, stride u) plus, at the stage's rate,
        the widest AMP block: per layer pair 6 (anti-aliased activation) + d (k-1)/2 + 6 + (k-1)/2 samples; then activation_post 6 and
        conv_post 3 samples.  Used by VQGAN.decode_stream to run the vocoder on a window and crop, bit-identically."""
        h = self.h
        frames, rate = 3.0, 1
        widest = 0
        for k, dils in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
            if h.resblock == "2":
                widest = max(widest, sum(6 + d * (k - 1) // 2 for d in dils))
            else:
                widest = max(widest, sum(12 + d * (k - 1) // 2 + (k - 1) // 2 for d in dils))
        for u in h.upsample_rates:
            frames += 1.0 / rate
            rate *= u
            frames += widest / rate
        frames += 9.0 / rate
        return int(frames) + 2

    def set_streams(self, n_streams: int) -> None:
        """1: all kernels on the current stream; 3 (default): the AMP blocks of a stage overlap on side streams."""
        with torch.cuda.device(self._device()):
            _lib.check(_lib.lib().dmel_bigvgan_set_streams(self.native(), n_streams), "bigvgan_set_streams")

    @torch.no_grad()
    def forward(self, x):
        """mel (B, num_mels, T) -> audio (B, 1, T * prod(upsample_rates))        (bigvgan.py:367-393)"""
        _lib.require_cuda(x, "mel")
        if x.ndim != 3 or x.shape[1] != self.h.num_mels:
            raise ValueError(f"expected (B, {self.h.num_mels}, T), got {tuple(x.shape)}")
        x = x.float().contiguous()
        B, _, T = x.shape
        up = 1
        for u in self.h.upsample_rates:
            up *= u
        L = _lib.lib()
        with torch.cuda.device(x.device):
            h = self.native()
            ws = self._ws.get(L.dmel_bigvgan_workspace_bytes(h, B, T), x.device)
        # through PyTorch's dispatcher (dmel_codec_amd/torch_ops.py): torch.ops.dmel_hip.bigvgan_forward -> dmel_bigvgan_forward
        return torch.ops.dmel_hip.bigvgan_forward(h, x, up, ws)
