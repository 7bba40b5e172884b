"""`torch.ops.dmel_hip.*`: the hot-path kernels registered with PyTorch's dispatcher (torch.library custom ops over the C ABI of
libdmel_hip.so), so that they are visible to the dispatcher, to autograd and to torch.compile the way the reference's one native op
is (`anti_alias_activation_cuda.forward`, alias_free_activation/cuda/anti_alias_activation.cpp:19-23, JIT-built by load.py:31-48).

    torch.ops.dmel_hip.anti_alias_activation_forward(input, up_filter, down_filter, alpha, beta)     # = fwd_cuda, same arguments
    torch.ops.dmel_hip.aa_snake(x, alpha, beta?, up_filter, down_filter, logscale)                    # + autograd (native backward)
    torch.ops.dmel_hip.conv1d_dilated(x, weight, bias?, dilation)                                    # + autograd (dgrad / wgrad kernels)
    torch.ops.dmel_hip.stft_logmel(audio, lengths?, sample_rate, n_fft, win_length, hop_length, n_mels, f_min, f_max)
    torch.ops.dmel_hip.stft_magnitude(audio, n_fft, win_length, hop_length)                           # + autograd (DFT-as-GEMM backward)
    torch.ops.dmel_hip.wavenet_forward(handle, x, condition?, in_lengths?, out_lengths?, group_repeat, out_channels)
    torch.ops.dmel_hip.bigvgan_forward(handle, mel, total_upsampling)

Module-level ops take the native handle (an integer, owned by the mirror module) -- weights live inside the handle in MFMA tile
order, not in tensors.  Every op has a fake (meta) implementation for shape propagation; none has a CPU implementation: the product
path fails loudly without the GPU library."""
from __future__ import annotations

import collections
import ctypes as C
import weakref
from typing import Optional

import torch
from torch import Tensor

from . import _lib


# ----------------------------------------------------------------------------------------------------- filters (host constants)
class _TensorKeyedCache:
    """Values derived from a tensor (packed-weight handles, host copies of filter taps), tied to the LIFETIME of that tensor.

    An entry is found through id(tensor) and is valid only while its weak reference still points to the very same object at the same
    `_version` (and, when a second tensor took part, while that one is the same object at the same version).  A data_ptr() key is not
    enough: the caching allocator hands the address of a freed weight to the next tensor of that size, and a fresh tensor starts at
    version 0 again.  When the tensor dies its entry is destroyed by a finaliser; beyond `capacity` the least recently used entry goes.
    """

    def __init__(self, capacity: int, destroy=None):
        self.capacity, self.destroy = capacity, destroy
        self.entries: "collections.OrderedDict" = collections.OrderedDict()

    def _drop(self, key):
        e = self.entries.pop(key, None)
        if e is not None and self.destroy is not None:
            self.destroy(e["value"])

    def get(self, tensor: Tensor, extra, other: Optional[Tensor] = None, any_other: bool = False):
        key = (id(tensor), extra)
        e = self.entries.get(key)
        if e is None:
            return None
        ok = e["ref"]() is tensor and e["version"] == tensor._version
        if ok and not any_other:
            o = e["other"]() if e["other"] is not None else None
            ok = (o is other) and (other is None or e["other_version"] == other._version)
        if not ok:
            self._drop(key)
            return None
        self.entries.move_to_end(key)
        return e["value"]

    def put(self, tensor: Tensor, extra, value, other: Optional[Tensor] = None):
        key = (id(tensor), extra)
        self._drop(key)
        while len(self.entries) >= self.capacity:
            self._drop(next(iter(self.entries)))
        self.entries[key] = dict(ref=weakref.ref(tensor), version=tensor._version, value=value,
                                 other=weakref.ref(other) if other is not None else None,
                                 other_version=other._version if other is not None else None)
        weakref.finalize(tensor, self._expire, key, weakref.ref(tensor))

    def _expire(self, key, ref):
        e = self.entries.get(key)
        if e is not None and e["ref"]() is None:       # still the dead tensor's entry (not a newer tensor that reuses the id)
            self._drop(key)


_taps_cache = _TensorKeyedCache(256)


def _host_taps(f: Tensor) -> Tensor:
    """12 filter taps as a host fp32 tensor.  The kernel takes them as launch constants; a device buffer is copied to the host ONCE per
    (tensor object, version) -- not per call (the filters are registered buffers that never change after load)."""
    if f.device.type == "cpu" and f.dtype == torch.float32 and f.is_contiguous():
        t = f.detach().reshape(-1)
    else:
        t = _taps_cache.get(f, None)
        if t is None:
            t = f.detach().to("cpu", torch.float32).contiguous().reshape(-1)
            _taps_cache.put(f, None, t)
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
_conv_cache = _TensorKeyedCache(64, destroy=lambda h: _lib.lib().dmel_conv_destroy(h))


def _conv_handle(weight: Tensor, bias: Optional[Tensor], dilation: int, for_backward: bool = False) -> int:
    """Packed-weight handle of (weight, bias, dilation): one entry per weight tensor, shared by forward and backward (backward does not
    read the bias, so it takes whatever handle the forward built), rebuilt when the weight or the bias is modified in place or replaced."""
    h = _conv_cache.get(weight, dilation, bias, any_other=for_backward)
    if h is None:
        Cout, Cin, k = weight.shape
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        hv = C.c_void_p()
        _lib.check(_lib.lib().dmel_conv_create(C.byref(hv), w.data_ptr(), _lib.ptr(b), Cout, Cin, k, dilation), "conv_create")
        h = hv.value
        _conv_cache.put(weight, dilation, h, bias)
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
        h = _conv_handle(weight, None, dilation, for_backward=True)
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
_convt_cache = _TensorKeyedCache(32, destroy=lambda h: _lib.lib().dmel_conv_transpose1d_destroy(h))


def _convt_handle(weight: Tensor, bias: Optional[Tensor], stride: int) -> int:
    h = _convt_cache.get(weight, stride, bias)
    if h is None:
        Cin, Cout, k = weight.shape
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = bias.detach().to("cpu", torch.float32).contiguous() if bias is not None else None
        hv = C.c_void_p()
        _lib.check(_lib.lib().dmel_conv_transpose1d_create(C.byref(hv), w.data_ptr(), _lib.ptr(b), Cin, Cout, k, stride), "conv_transpose1d_create")
        h = hv.value
        _convt_cache.put(weight, stride, h, bias)
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


_stft_grads: dict = {}


def _stft_grad_handle(device, n_fft, win_length, hop_length) -> int:
    key = (str(device), n_fft, win_length, hop_length)
    h = _stft_grads.get(key)
    if h is None:
        hv = C.c_void_p()
        window = torch.hann_window(win_length, dtype=torch.float32)           # utils/spectrogram.py:53
        with torch.cuda.device(device):
            _lib.check(_lib.lib().dmel_stft_grad_create(C.byref(hv), n_fft, win_length, hop_length, window.data_ptr()), "stft_grad_create")
        h = hv.value
        _stft_grads[key] = h
    return h


@torch.library.custom_op("dmel_hip::stft_magnitude_backward", mutates_args=(), device_types="cuda")
def stft_magnitude_backward(audio: Tensor, grad: Tensor, n_fft: int, win_length: int, hop_length: int) -> Tensor:
    """d loss / d audio from d loss / d |S| (B, L // hop, n_fft // 2 + 1): dmel_stft_magnitude_backward_f32 -- both DFTs as GEMMs on the
    library's convolution kernel, overlap-add with the reflect padding folded back."""
    _lib.require_cuda(audio, "audio")
    y = audio.float()
    if y.stride(-1) != 1:
        y = y.contiguous()
    g = grad.float().contiguous()
    B, Ls = y.shape
    L = _lib.lib()
    dy = torch.empty(B, Ls, dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        h = _stft_grad_handle(y.device, n_fft, win_length, hop_length)
        ws = torch.empty(L.dmel_stft_grad_workspace_bytes(h, B, Ls), dtype=torch.uint8, device=y.device)
        _lib.check(L.dmel_stft_magnitude_backward_f32(h, y.data_ptr(), y.stride(0), g.data_ptr(), dy.data_ptr(), dy.stride(0), B, Ls,
                                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "stft_magnitude_backward")
    return dy


@stft_magnitude_backward.register_fake
def _(audio, grad, n_fft, win_length, hop_length):
    return audio.new_empty(audio.shape, dtype=torch.float32)


def _stft_mag_setup(ctx, inputs, output):
    audio, n_fft, win_length, hop_length = inputs
    ctx.save_for_backward(audio)
    ctx.cfg = (n_fft, win_length, hop_length)


def _stft_mag_backward(ctx, g):
    (audio,) = ctx.saved_tensors
    return stft_magnitude_backward(audio, g, *ctx.cfg), None, None, None


stft_magnitude.register_autograd(_stft_mag_backward, setup_context=_stft_mag_setup)


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
