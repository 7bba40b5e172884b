"""WaveNet on the MI355X.  Drop-in for dmel_codec/models/modules/wavenet.py (reference): same class names, ctor
kwargs, parameter names (state-dict keys) and initialisation; forward runs the fused HIP path
(csrc/modules.hip: dmel_wavenet_forward) -- one implicit-GEMM launch per gated conv and one per output projection.
When gradients are required (grad mode on and a parameter or input requires grad) forward runs the native training path
instead (dmel_wavenet_forward_train / dmel_wavenet_backward behind a torch.autograd.Function): the same arithmetic unfused,
with hand-written backward kernels -- what the reference gets from autograd (codec_lit_modules.py:236,315)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import nn

from ... import _lib, torch_ops  # noqa: F401  (torch_ops registers torch.ops.dmel_hip.*)
from ._native import NativeModule


class LinearNorm(nn.Module):
    """wavenet.py:31-44 (parameter container; only used by the unused diffusion branch)."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias)
        nn.init.xavier_uniform_(self.linear.weight)
        if bias:
            nn.init.constant_(self.linear.bias, 0.0)


class ConvNorm(nn.Module):
    """wavenet.py:47-81 (parameter container)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain="linear"):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.kaiming_normal_(self.conv.weight)


class ResidualBlock(nn.Module):
    """wavenet.py:84-135 (parameter container; the arithmetic of :116-135 runs inside dmel_wavenet_forward)."""

    def __init__(self, residual_channels, use_linear_bias=False, dilation=1, condition_channels=None):
        super().__init__()
        self.conv_layer = ConvNorm(residual_channels, 2 * residual_channels, kernel_size=3, stride=1,
                                   padding=dilation, dilation=dilation)
        if condition_channels is not None:
            # dead parameters in the reference too (never used when diffusion_step is None, wavenet.py:119-121)
            self.diffusion_projection = LinearNorm(residual_channels, residual_channels, use_linear_bias)
            self.condition_projection = ConvNorm(condition_channels, 2 * residual_channels, kernel_size=1)
        self.output_projection = ConvNorm(residual_channels, 2 * residual_channels, kernel_size=1)


class _WaveNetTrainFn(torch.autograd.Function):
    """Native forward-with-saved-activations / backward of the whole WaveNet (include/dmel_hip.h, training path)."""

    @staticmethod
    def forward(ctx, module, x, condition, *params):
        L = _lib.lib()
        N, _, T = x.shape
        dev = x.device
        y = torch.empty(N, module.output_channels, T, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            h = module.native()
            ws = torch.empty(L.dmel_wavenet_train_workspace_bytes(h, N, T), dtype=torch.uint8, device=dev)
            _lib.check(L.dmel_wavenet_forward_train(h, x.data_ptr(), _lib.ptr(condition), y.data_ptr(), N, T, ws.data_ptr(),
                                                    ws.numel(), _lib.stream_ptr()), "wavenet_forward_train")
        ctx.module, ctx.handle, ctx.ws = module, h, ws
        module._begin_train_call(ctx)
        ctx.save_for_backward(x, condition if condition is not None else torch.empty(0, device=dev))
        ctx.has_cond = condition is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        module, h, ws = ctx.module, ctx.handle, ctx.ws
        x, cond = ctx.saved_tensors
        cond = cond if ctx.has_cond else None
        module._check_train_call(ctx)
        L = _lib.lib()
        N, _, T = x.shape
        dev = x.device
        dy = dy.float().contiguous()
        need_dx, need_dc = ctx.needs_input_grad[1], ctx.has_cond and ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_dx else None
        dc = torch.empty_like(cond) if need_dc else None
        trained = module._trained_parameters()
        needs = ctx.needs_input_grad[3:]
        # data-parallel training: while a ddp.GradReducer is armed, every block's slice of the flat gradient buffer leaves for its
        # all-reduce from INSIDE the native backward (on_ready fires when the block's last kernel is enqueued), so the exchange of
        # block k overlaps the backward of blocks k-1 ... 0
        stream_out = module._can_stream_grads([p for _, p in trained], needs)
        failure = []
        with torch.cuda.device(dev):
            flat = torch.empty(L.dmel_wavenet_grad_floats(h), dtype=torch.float32, device=dev)
            hook = None
            if stream_out:
                sink = module._grad_sink

                def on_ready(_user, offset, numel):
                    try:
                        sink.submit(flat[offset:offset + numel])
                    except BaseException as e:      # never unwind through the C frame
                        failure.append(e)
                hook = _lib.GRAD_READY_FN(on_ready)
            _lib.check(L.dmel_wavenet_backward_hooked(h, x.data_ptr(), _lib.ptr(cond), dy.data_ptr(), _lib.ptr(dx), _lib.ptr(dc),
                                                      flat.data_ptr(), N, T, ws.data_ptr(), ws.numel(), _lib.stream_ptr(),
                                                      hook if hook is not None else _lib.GRAD_READY_FN(), None),
                       "wavenet_backward")
        if failure:
            raise failure[0]
        slots = []
        off, num = C.c_int64(), C.c_int64()
        for key, prm in trained:
            _lib.check(L.dmel_wavenet_grad_slot(h, key.encode(), C.byref(off), C.byref(num)), "wavenet_grad_slot")
            slots.append((prm, off.value, num.value))
        grads = module._deliver_grads(flat, slots, needs, streamed=stream_out)
        return (None, dx, dc, *grads)


class WaveNet(NativeModule):
    """wavenet.py:138-225."""

    _destroy_symbol = "dmel_wavenet_destroy"
    _set_symbol = "dmel_wavenet_set_tensor"
    _finalize_symbol = "dmel_wavenet_finalize"
    _train_precision_symbol = "dmel_wavenet_set_train_precision"
    _precision_symbol = "dmel_wavenet_set_precision"
    _refresh_symbol = "dmel_wavenet_refresh"

    def __init__(self, input_channels: Optional[int] = None, output_channels: Optional[int] = None,
                 residual_channels: int = 512, residual_layers: int = 20, dilation_cycle: Optional[int] = 4,
                 is_diffusion: bool = False, condition_channels: Optional[int] = None):
        super().__init__()
        if is_diffusion:
            raise NotImplementedError("is_diffusion=True is never set by a reference config and is not built")
        self.input_projection = None
        if input_channels is not None and input_channels != residual_channels:
            self.input_projection = ConvNorm(input_channels, residual_channels, kernel_size=1)
        if input_channels is None:
            input_channels = residual_channels
        self.input_channels = input_channels
        self.residual_channels = residual_channels
        self.dilation_cycle = dilation_cycle
        self.condition_channels = condition_channels
        self.residual_layers = nn.ModuleList([
            ResidualBlock(residual_channels=residual_channels, use_linear_bias=False,
                          dilation=2 ** (i % dilation_cycle) if dilation_cycle else 1,
                          condition_channels=condition_channels)
            for i in range(residual_layers)])
        self.skip_projection = ConvNorm(residual_channels, residual_channels, kernel_size=1)
        self.output_projection = None
        self.output_channels = residual_channels
        if output_channels is not None and output_channels != residual_channels:
            self.output_projection = ConvNorm(residual_channels, output_channels, kernel_size=1)
            self.output_channels = output_channels
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (nn.Conv1d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias, 0)

    def _native_state(self) -> dict:
        return {k: v for k, v in self.state_dict().items() if "diffusion_projection" not in k}

    def _create_native(self) -> int:
        h = C.c_void_p()
        _lib.check(_lib.lib().dmel_wavenet_create(C.byref(h), self.input_channels, self.output_channels,
                                                  self.residual_channels, len(self.residual_layers),
                                                  self.dilation_cycle or 0, self.condition_channels or 0),
                   "wavenet_create")
        if getattr(self, "_want_train", False):
            _lib.check(_lib.lib().dmel_wavenet_enable_training(h, 1), "wavenet_enable_training")
        return h.value

    def _trained_parameters(self):
        """(state-dict key, parameter) of everything the native backward produces a gradient for, in a fixed order
        (the dead diffusion_projection weights receive none, as in the reference)."""
        return [(k, p) for k, p in self.named_parameters() if "diffusion_projection" not in k]

    def forward(self, x, t=None, condition=None, in_lengths=None, out_lengths=None, group_repeat: int = 1):
        if torch.is_grad_enabled() and (x.requires_grad or (condition is not None and condition.requires_grad)
                                        or any(p.requires_grad for _, p in self._trained_parameters())):
            return self._forward_train(x, t, condition, in_lengths, out_lengths, group_repeat)
        with torch.no_grad():
            return self._forward_infer(x, t, condition, in_lengths, out_lengths, group_repeat)

    def _forward_train(self, x, t, condition, in_lengths, out_lengths, group_repeat):
        """Differentiable forward: masks (if given) are applied as plain tensor ops around the native call, exactly where
        the reference multiplies them (codec_lit_modules.py:189-190, 205-211)."""
        if t is not None:
            raise NotImplementedError("diffusion step input is not built (never used by the codec path)")
        _lib.require_cuda(x, "x")
        if x.ndim != 3 or x.shape[1] != self.input_channels:
            raise ValueError(f"expected (N, {self.input_channels}, T), got {tuple(x.shape)}")
        if (condition is not None) != bool(self.condition_channels):
            raise ValueError("condition tensor does not match condition_channels")
        N, _, T = x.shape
        if condition is not None:
            # the C side receives raw pointers and (N, T) only: a mis-shaped or mis-placed condition would be read out of bounds
            _lib.require_cuda(condition, "condition")
            if condition.device != x.device:
                raise ValueError(f"condition lives on {condition.device}, x on {x.device}")
            if condition.shape != (N, self.condition_channels, T):
                raise ValueError(f"condition must be {(N, self.condition_channels, T)}, got {tuple(condition.shape)}")
        if any(p.device != x.device for _, p in self._trained_parameters()):
            raise ValueError(f"WaveNet parameters and input live on different devices (input on {x.device})")

        def mask(v):
            v = v.reshape(-1).to(device=x.device, dtype=torch.int64)
            if v.numel() * group_repeat != N:
                raise ValueError("lengths do not match the batch")
            m = (torch.arange(T, device=x.device)[None, :] < v[:, None]).to(torch.float32)
            return m.repeat_interleave(group_repeat, dim=0)[:, None, :]

        x = x.float()
        if in_lengths is not None:
            x = x * mask(in_lengths)
        if not getattr(self, "_want_train", False):
            self._want_train = True
            self._free_native()              # the training images are packed at finalize: rebuild the handle once
        params = [p for _, p in self._trained_parameters()]
        y = _WaveNetTrainFn.apply(self, x.contiguous(), condition.float().contiguous() if condition is not None else None, *params)
        if out_lengths is not None:
            y = y * mask(out_lengths)
        return y

    def _forward_infer(self, x, t=None, condition=None, in_lengths=None, out_lengths=None, group_repeat: int = 1):
        """x (N, Cin, T), condition (N, Ccond, T) -> (N, Cout, T).
        Extensions (optional): in_lengths / out_lengths (N // group_repeat,) int64 fuse the `x * mask` in front of
        and behind the stack (codec_lit_modules.py:471-477, 505-506)."""
        if t is not None:
            raise NotImplementedError("diffusion step input is not built (never used by the codec path)")
        _lib.require_cuda(x, "x")
        if x.ndim != 3 or x.shape[1] != self.input_channels:
            raise ValueError(f"expected (N, {self.input_channels}, T), got {tuple(x.shape)}")
        x = x.float().contiguous()
        N, _, T = x.shape
        if (condition is not None) != bool(self.condition_channels):
            raise ValueError("condition tensor does not match condition_channels")
        if condition is not None:
            _lib.require_cuda(condition, "condition")
            if condition.shape != (N, self.condition_channels, T):
                raise ValueError(f"condition must be {(N, self.condition_channels, T)}, got {tuple(condition.shape)}")
            condition = condition.float().contiguous()
        dev = x.device

        def lens(v):
            if v is None:
                return None
            v = v.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
            if v.numel() * group_repeat != N:
                raise ValueError("lengths do not match the batch")
            return v

        il, ol = lens(in_lengths), lens(out_lengths)
        L = _lib.lib()
        with torch.cuda.device(dev):      # handle creation (weight upload, side streams) must happen on x's device
            h = self.native()
            ws = self._ws.get(L.dmel_wavenet_workspace_bytes(h, N, T), dev)
        # through PyTorch's dispatcher (dmel_codec_amd/torch_ops.py): torch.ops.dmel_hip.wavenet_forward -> dmel_wavenet_forward
        return torch.ops.dmel_hip.wavenet_forward(h, x, condition, il, ol, group_repeat, self.output_channels, ws)
