"""`torch.ops.dmel_hip.*`: the hot-path kernels registered with PyTorch's dispatcher (torch.library custom ops over the C ABI of
libdmel_hip.so), so that they are visible to the dispatcher, to autograd and to torch.compile the way the reference's one native op
is (`anti_alias_activation_cuda.forward`, alias_free_activation/cuda/anti_alias_activation.cpp:19-23, JIT-built by load.py:31-48).

    torch.ops.dmel_hip.anti_alias_activation_forward(input, up_filter, down_filter, alpha, beta)     # = fwd_cuda, same arguments
    torch.ops.dmel_hip.aa_snake(x, alpha, beta?, up_filter, down_filter, logscale)                    # + autograd (native backward)
    torch.ops.dmel_hip.conv1d_dilated(x, weight, bias?, dilation)                                    # + autograd (dgrad / wgrad kernels)
    torch.ops.dmel_hip.stft_logmel(audio, lengths?, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max)
    torch.ops.dmel_hip.wavenet_forward(handle, x, condition?, in_lengths?, out_lengths?, group_repeat, out_channels)
    torch.ops.dmel_hip.bigvgan_forward(handle, mel, total_upsampling)

Module-level ops take the native handle (an integer, owned by the mirror module) -- weights live inside the handle in MFMA tile
order, not in tensors.  Every op has a fake (meta) implementation for shape propagation; none has a CPU implementation: the product
path fails loudly without the GPU library."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor

from . import _lib


# ----------------------------------------------------------------------------------------------------- filters (host constants)
_taps_cache: dict = {}


def _host_taps(f: Tensor) -> Tensor:
    """12 filter taps as a host fp32 tensor.  The kernel takes them as launch constants; a device buffer is copied to the host ONCE per
    (storage, version) -- not per call (the filters are registered buffers that never change after load)."""
    if f.device.type == "cpu" and f.dtype == torch.float32 and f.is_contiguous():
        t = f.detach().reshape(-1)
    else:
        key = (f.data_ptr(), f._version, str(f.device))
        t = _taps_cache.get(key)
        if t is None:
            if len(_taps_cache) > 256:
                _taps_cache.clear()
            t = f.detach().to("cpu", torch.float32).contiguous().reshape(-1)
            _taps_cache[key] = t
    if t.numel() != 12:
        raise NotImplementedError("the fused anti-alias kernel is built for 12-tap filters (the only setting BigVGAN uses)")
    return t


# ----------------------------------------------------------------------------------------------------- anti-aliased snake
@torch.library.custom_op("dmel_hip::aa_snake", mutates_args=(), device_types="cuda")
def aa_snake(x: Tensor, alpha: Tensor, beta: Optional[Tensor], up_filter: Tensor, down_filter: Tensor, logscale: bool) -> Tensor:
    _lib.require_cuda(x, "x")
    x = x.float().contiguous()
    B, Cc, T = x.shape
    y = torch.empty_like(x)
    up, dn = _host_taps(up_filter), _host_taps(down_filter)
    a = alpha.detach().to(x.device, torch.float32).contiguous()
    b = beta.detach().to(x.device, torch.float32).contiguous() if beta is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dmel_aa_snake_f32(x.data_ptr(), y.data_ptr(), a.data_ptr(), _lib.ptr(b), up.data_ptr(), dn.data_ptr(),
                                                int(logscale), B, Cc, T, _lib.stream_ptr()), "aa_snake")
    return y


@aa_snake.register_fake
def _(x, alpha, beta, up_filter, down_filter, logscale):
    return torch.empty_like(x, dtype=torch.float32)


@torch.library.custom_op("dmel_hip::aa_snake_backward", mutates_args=(), device_types="cuda")
def aa_snake_backward(x: Tensor, dy: Tensor, alpha: Tensor, beta: Optional[Tensor], up_filter: Tensor, down_filter: Tensor,
                      logscale: bool) -> tuple[Tensor, Tensor, Tensor]:
    x = x.float().contiguous()
    dy = dy.float().contiguous()
    B, Cc, T = x.shape
    a = alpha.detach().to(x.device, torch.float32).contiguous()
    b = beta.detach().to(x.device, torch.float32).contiguous() if beta is not None else None
    dx, da = torch.empty_like(x), torch.empty_like(a)
    db = torch.empty_like(b) if b is not None else None
    up, dn = _host_taps(up_filter), _host_taps(down_filter)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dmel_aa_snake_backward_f32(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), a.data_ptr(), _lib.ptr(b), da.data_ptr(),
                                                         _lib.ptr(db), up.data_ptr(), dn.data_ptr(), int(logscale), B, Cc, T,
                                                         _lib.stream_ptr()), "aa_snake_backward")
    return dx, da, db if db is not None else torch.empty(0, device=x.device)


@aa_snake_backward.register_fake
def _(x, dy, alpha, beta, up_filter, down_filter, logscale):
    return (torch.empty_like(x, dtype=torch.float32), torch.empty_like(alpha, dtype=torch.float32),
            torch.empty_like(beta, dtype=torch.float32) if beta is not None else x.new_empty(0))


def _aa_setup(ctx, inputs, output):
    x, alpha, beta, up_filter, down_filter, logscale = inputs
    ctx.save_for_backward(x, alpha, beta if beta is not None else x.new_empty(0), up_filter, down_filter)
    ctx.has_beta, ctx.logscale = beta is not None, logscale


def _aa_backward(ctx, dy):
    x, alpha, beta, up_filter, down_filter = ctx.saved_tensors
    dx, da, db = aa_snake_backward(x, dy, alpha, beta if ctx.has_beta else None, up_filter, down_filter, ctx.logscale)
    return dx, da, (db if ctx.has_beta else None), None, None, None


aa_snake.register_autograd(_aa_backward, setup_context=_aa_setup)


@torch.library.custom_op("dmel_hip::anti_alias_activation_forward", mutates_args=(), device_types="cuda")
def anti_alias_activation_forward(input: Tensor, up_filter: Tensor, down_filter: Tensor, alpha: Tensor, beta: Tensor) -> Tensor:
    """fwd_cuda(input, up_filter, down_filter, alpha, beta) of the reference (anti_alias_activation_cuda.cu:212-246): alpha / beta are
    LOG-scale (the kernel applies exp, :87-89), forward only, output allocated here."""
    return aa_snake(input, alpha, beta, up_filter, down_filter, True)


@anti_alias_activation_forward.register_fake
def _(input, up_filter, down_filter, alpha, beta):
    return torch.empty_like(input, dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------- dilated conv1d
_conv_cache: dict = {}


def _conv_handle(weight: Tensor, bias: Optional[Tensor], dilation: int) -> int:
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), dilation,
           None if bias is None else (bias.data_ptr(), bias._version))
    h = _conv_cache.get(key)
    if h is None:
        if len(_conv_cache) >= 64:
            for old in _conv_cache.values():
                _lib.lib().dmel_conv_destroy(old)
            _conv_cache.clear()
        Cout, Cin, k = weight.shape
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        hv = C.c_void_p()
        _lib.check(_lib.lib().dmel_conv_create(C.byref(hv), w.data_ptr(), _lib.ptr(b), Cout, Cin, k, dilation), "conv_create")
        h = hv.value
        _conv_cache[key] = h
    return h


@torch.library.custom_op("dmel_hip::conv1d_dilated", mutates_args=(), device_types="cuda")
def conv1d_dilated(x: Tensor, weight: Tensor, bias: Optional[Tensor], dilation: int) -> Tensor:
    """F.conv1d(x, weight, bias, padding=dilation * (k - 1) // 2, dilation=dilation) for odd k ("same" length), on the split-fp32
    implicit-GEMM kernel.  The packed weight image is cached per (weight storage, version)."""
    _lib.require_cuda(x, "x")
    x = x.float().contiguous()
    B, Cin, T = x.shape
    if weight.ndim != 3 or weight.shape[1] != Cin or weight.shape[2] % 2 != 1:
        raise ValueError(f"weight must be (Cout, {Cin}, odd k), got {tuple(weight.shape)}")
    y = torch.empty(B, weight.shape[0], T, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        h = _conv_handle(weight, bias, dilation)
        _lib.check(_lib.lib().dmel_conv_forward(h, x.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()), "conv_forward")
    return y


@conv1d_dilated.register_fake
def _(x, weight, bias, dilation):
    return x.new_empty((x.shape[0], weight.shape[0], x.shape[2]), dtype=torch.float32)


@torch.library.custom_op("dmel_hip::conv1d_dilated_backward", mutates_args=(), device_types="cuda")
def conv1d_dilated_backward(x: Tensor, dy: Tensor, weight: Tensor, dilation: int) -> tuple[Tensor, Tensor, Tensor]:
    x = x.float().contiguous()
    dy = dy.float().contiguous()
    B, Cin, T = x.shape
    dx = torch.empty_like(x)
    dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
    db = torch.empty(weight.shape[0], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        h = _conv_handle(weight, None, dilation)
        L = _lib.lib()
        _lib.check(L.dmel_conv_backward_data(h, dy.data_ptr(), dx.data_ptr(), B, T, _lib.stream_ptr()), "conv_backward_data")
        _lib.check(L.dmel_conv_backward_weight(h, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), B, T, _lib.stream_ptr()),
                   "conv_backward_weight")
    return dx, dw, db


@conv1d_dilated_backward.register_fake
def _(x, dy, weight, dilation):
    return (torch.empty_like(x, dtype=torch.float32), torch.empty_like(weight, dtype=torch.float32),
            x.new_empty((weight.shape[0],), dtype=torch.float32))


def _conv_setup(ctx, inputs, output):
    x, weight, bias, dilation = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias, ctx.dilation = bias is not None, dilation


def _conv_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw, db = conv1d_dilated_backward(x, dy, weight, ctx.dilation)
    return dx, dw, (db if ctx.has_bias else None), None


conv1d_dilated.register_autograd(_conv_backward, setup_context=_conv_setup)


# ----------------------------------------------------------------------------------------------------- transposed conv / output conv
_convt_cache: dict = {}


def _convt_handle(weight: Tensor, bias: Optional[Tensor], stride: int) -> int:
    key = (weight.data_ptr(), weight._version, str(weight.device), tuple(weight.shape), stride,
           None if bias is None else (bias.data_ptr(), bias._version))
    h = _convt_cache.get(key)
    if h is None:
        if len(_convt_cache) >= 32:
            for old in _convt_cache.values():
                _lib.lib().dmel_conv_transpose1d_destroy(old)
            _convt_cache.clear()
        Cin, Cout, k = weight.shape
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        hv = C.c_void_p()
        _lib.check(_lib.lib().dmel_conv_transpose1d_create(C.byref(hv), w.data_ptr(), _lib.ptr(b), Cin, Cout, k, stride), "conv_transpose1d_create")
        h = hv.value
        _convt_cache[key] = h
    return h


@torch.library.custom_op("dmel_hip::conv_transpose1d", mutates_args=(), device_types="cuda")
def conv_transpose1d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int) -> Tensor:
    """F.conv_transpose1d(x, weight, bias, stride=stride, padding=(k - stride) // 2) for k == 2 * stride (every BigVGAN up-sampler,
    bigvgan.py:320-334): `stride` phase sub-convolutions on the implicit-GEMM kernel.  weight (Cin, Cout, k), weight norm folded."""
    _lib.require_cuda(x, "x")
    x = x.float().contiguous()
    B, Cin, T = x.shape
    if weight.ndim != 3 or weight.shape[0] != Cin:
        raise ValueError(f"weight must be ({Cin}, Cout, k), got {tuple(weight.shape)}")
    y = torch.empty(B, weight.shape[1], T * stride, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        h = _convt_handle(weight, bias, stride)
        _lib.check(_lib.lib().dmel_conv_transpose1d_forward(h, x.data_ptr(), y.data_ptr(), B, T, _lib.stream_ptr()), "conv_transpose1d_forward")
    return y


@conv_transpose1d.register_fake
def _(x, weight, bias, stride):
    return x.new_empty((x.shape[0], weight.shape[1], x.shape[2] * stride), dtype=torch.float32)


@torch.library.custom_op("dmel_hip::conv_post", mutates_args=(), device_types="cuda")
def conv_post(x: Tensor, weight: Tensor, bias: float, activation: str) -> Tensor:
    """The C -> 1 convolution that ends the vocoder (bigvgan.py:386-391): act(F.conv1d(x, weight (1, C, K), padding=K // 2) + bias),
    activation "none" | "tanh" | "clamp"."""
    _lib.require_cuda(x, "x")
    x = x.float().contiguous()
    B, Cc, T = x.shape
    if weight.ndim != 3 or weight.shape[0] != 1 or weight.shape[1] != Cc or weight.shape[2] % 2 != 1:
        raise ValueError(f"weight must be (1, {Cc}, odd K), got {tuple(weight.shape)}")
    act = {"none": 0, "tanh": 2, "clamp": 3}[activation]
    w = weight.detach().to(x.device, torch.float32).contiguous()
    y = torch.empty(B, 1, T, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().dmel_conv_post_f32(x.data_ptr(), w.data_ptr(), float(bias), act, y.data_ptr(), B, Cc, weight.shape[2], T,
                                                 _lib.stream_ptr()), "conv_post")
    return y


@conv_post.register_fake
def _(x, weight, bias, activation):
    return x.new_empty((x.shape[0], 1, x.shape[2]), dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------- STFT -> log-mel
_plans: dict = {}


def _stft_plan(device, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max) -> int:
    key = (str(device), sample_rate, n_fft, win_length, hop_length, n_mels, float(f_min), float(f_max))
    h = _plans.get(key)
    if h is None:
        hv = C.c_void_p()
        window = torch.hann_window(win_length, dtype=torch.float32)           # utils/spectrogram.py:53
        with torch.cuda.device(device):
            _lib.check(_lib.lib().dmel_stft_plan_create(C.byref(hv), sample_rate, n_fft, win_length, hop_length, n_mels, float(f_min),
                                                        float(f_max), window.data_ptr()), "stft_plan_create")
        h = hv.value
        _plans[key] = h
    return h


@torch.library.custom_op("dmel_hip::stft_logmel", mutates_args=(), device_types="cuda")
def stft_logmel(audio: Tensor, lengths: Optional[Tensor], sample_rate: int, n_fft: int, win_length: int, hop_length: int, n_mels: int,
                f_min: float, f_max: float) -> Tensor:
    """LinearSpectrogram.forward of the reference (utils/spectrogram.py:41-81) in one launch: audio (B, L) fp32 -> (B, n_mels, L // hop).
    f_max = 0 means sample_rate / 2.  lengths (optional, (B,) int64 samples): frames at or behind lengths // hop are written as 0."""
    _lib.require_cuda(audio, "audio")
    y = audio.float()
    if y.ndim != 2:
        raise ValueError(f"expected (B, L), got {tuple(y.shape)}")
    if y.stride(-1) != 1:
        y = y.contiguous()
    B, Ls = y.shape
    lens = lengths.reshape(-1).to(device=y.device, dtype=torch.int64).contiguous() if lengths is not None else None
    L = _lib.lib()
    with torch.cuda.device(y.device):
        plan = _stft_plan(y.device, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max)
        T = L.dmel_stft_num_frames(plan, Ls)
        out = torch.empty(B, n_mels, T, dtype=torch.float32, device=y.device)
        _lib.check(L.dmel_stft_logmel_f32(plan, y.data_ptr(), y.stride(0), _lib.ptr(lens), out.data_ptr(), B, Ls, _lib.stream_ptr()),
                   "stft_logmel")
    return out


@stft_logmel.register_fake
def _(audio, lengths, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max):
    return audio.new_empty((audio.shape[0], n_mels, audio.shape[1] // hop_length), dtype=torch.float32)


@torch.library.custom_op("dmel_hip::stft_magnitude", mutates_args=(), device_types="cuda")
def stft_magnitude(audio: Tensor, n_fft: int, win_length: int, hop_length: int) -> Tensor:
    """Linear STFT magnitudes with the reference's framing (reflect pad (n_fft - hop) / 2, periodic hann of win_length centred in
    n_fft, center=False, sqrt(re^2 + im^2 + 1e-9); utils/spectrogram.py:58-76): audio (B, L) -> (B, L // hop, n_fft // 2 + 1),
    frame-major.  Same kernel as stft_logmel with the mel stage skipped (dmel_stft_f32)."""
    _lib.require_cuda(audio, "audio")
    y = audio.float()
    if y.ndim != 2:
        raise ValueError(f"expected (B, L), got {tuple(y.shape)}")
    if y.stride(-1) != 1:
        y = y.contiguous()
    B, Ls = y.shape
    L = _lib.lib()
    with torch.cuda.device(y.device):
        plan = _stft_plan(y.device, 16000, n_fft, win_length, hop_length, 1, 0.0, 0.0)       # the mel tables are not used
        T = L.dmel_stft_num_frames(plan, Ls)
        out = torch.empty(B, T, n_fft // 2 + 1, dtype=torch.float32, device=y.device)
        _lib.check(L.dmel_stft_f32(plan, y.data_ptr(), y.stride(0), None, None, out.data_ptr(), B, Ls, _lib.stream_ptr()), "stft_magnitude")
    return out


@stft_magnitude.register_fake
def _(audio, n_fft, win_length, hop_length):
    return audio.new_empty((audio.shape[0], audio.shape[1] // hop_length, n_fft // 2 + 1), dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------- module-level ops
@torch.library.custom_op("dmel_hip::wavenet_forward", mutates_args=("workspace",), device_types="cuda")
def wavenet_forward(handle: int, x: Tensor, condition: Optional[Tensor], in_lengths: Optional[Tensor], out_lengths: Optional[Tensor],
                    group_repeat: int, out_channels: int, workspace: Tensor) -> Tensor:
    """WaveNet.forward (models/modules/wavenet.py:204-225) on a finalized dmel_wavenet handle: x (N, Cin, T) fp32 contiguous,
    condition (N, Ccond, T) or None, lengths (N // group_repeat,) int64 or None -> (N, out_channels, T).  workspace: caller-owned
    uint8 scratch of at least dmel_wavenet_workspace_bytes(handle, N, T) bytes (the library never allocates activations)."""
    N, _, T = x.shape
    L = _lib.lib()
    y = torch.empty(N, out_channels, T, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        ws = workspace
        _lib.check(L.dmel_wavenet_forward(handle, x.data_ptr(), _lib.ptr(condition), y.data_ptr(), N, T, _lib.ptr(in_lengths),
                                          _lib.ptr(out_lengths), group_repeat, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                   "wavenet_forward")
    return y


@wavenet_forward.register_fake
def _(handle, x, condition, in_lengths, out_lengths, group_repeat, out_channels, workspace):
    return x.new_empty((x.shape[0], out_channels, x.shape[2]), dtype=torch.float32)


@torch.library.custom_op("dmel_hip::bigvgan_forward", mutates_args=("workspace",), device_types="cuda")
def bigvgan_forward(handle: int, mel: Tensor, total_upsampling: int, workspace: Tensor) -> Tensor:
    """BigVGAN.forward (models/modules/bigvgan/bigvgan.py:367-393) on a finalized dmel_bigvgan handle: mel (B, n_mels, T) -> (B, 1, T * up).
    workspace: caller-owned uint8 scratch of at least dmel_bigvgan_workspace_bytes(handle, B, T) bytes."""
    B, _, T = mel.shape
    L = _lib.lib()
    y = torch.empty(B, 1, T * total_upsampling, dtype=torch.float32, device=mel.device)
    with torch.cuda.device(mel.device):
        ws = workspace
        _lib.check(L.dmel_bigvgan_forward(handle, mel.data_ptr(), y.data_ptr(), B, T, ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                   "bigvgan_forward")
    return y


@bigvgan_forward.register_fake
def _(handle, mel, total_upsampling, workspace):
    return mel.new_empty((mel.shape[0], 1, mel.shape[2] * total_upsampling), dtype=torch.float32)
