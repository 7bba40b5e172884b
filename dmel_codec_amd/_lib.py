"""ctypes binding of libdmel_hip.so (the C ABI declared in include/dmel_hip.h).

The product path has no fallback: if the library is missing this module raises, loudly.
torch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DMEL_LIB") or os.path.join(_HERE, "libdmel_hip.so")   # DMEL_LIB: A/B builds in tools/
_lib: Optional[C.CDLL] = None

i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p
GRAD_READY_FN = C.CFUNCTYPE(None, vp, C.c_int64, C.c_int64)     # dmel_grad_ready_fn (include/dmel_hip.h)


class BigVGANConfig(C.Structure):
    _fields_ = [
        ("num_mels", C.c_int),
        ("upsample_initial_channel", C.c_int),
        ("num_upsamples", C.c_int),
        ("upsample_rates", C.c_int * 8),
        ("upsample_kernel_sizes", C.c_int * 8),
        ("num_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * 8),
        ("resblock_dilations", (C.c_int * 3) * 8),
        ("snake_logscale", C.c_int),
        ("activation_snake", C.c_int),
        ("use_tanh_at_final", C.c_int),
        ("use_bias_at_final", C.c_int),
        ("resblock_type", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/dmel_hip.h declares
PROTOTYPES = {
    "dmel_last_error": (C.c_char_p, []),
    "dmel_abi_version": (C.c_int, []),
    "dmel_stft_plan_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]),
    "dmel_stft_plan_destroy": (None, [vp]),
    "dmel_mel_basis_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp]),
    "dmel_stft_plan_mel_basis": (C.c_int, [vp, vp]),
    "dmel_stft_num_frames": (C.c_int64, [vp, C.c_int64]),
    "dmel_stft_logmel_f32": (C.c_int, [vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_resample_f32": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, vp]),
    "dmel_stft_f32": (C.c_int, [vp, vp, C.c_int64, vp, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_aa_snake_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    "dmel_aa_snake_backward_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    "dmel_discriminator_create": (C.c_int, [C.POINTER(vp)]),
    "dmel_discriminator_destroy": (None, [vp]),
    "dmel_discriminator_set_tensor": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int]),
    "dmel_discriminator_finalize": (C.c_int, [vp]),
    "dmel_discriminator_out_frames": (C.c_int64, [vp, C.c_int64]),
    "dmel_discriminator_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int64]),
    "dmel_discriminator_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_discriminator_enable_training": (C.c_int, [vp, C.c_int]),
    "dmel_discriminator_set_train_precision": (C.c_int, [vp, C.c_int]),
    "dmel_discriminator_refresh": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp), vp]),
    "dmel_discriminator_train_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int64]),
    "dmel_discriminator_grad_floats": (C.c_int64, [vp]),
    "dmel_discriminator_grad_slot": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dmel_discriminator_forward_train": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_discriminator_backward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_convnext_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "dmel_convnext_destroy": (None, [vp]),
    "dmel_convnext_set_tensor": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), C.c_int]),
    "dmel_convnext_enable_training": (C.c_int, [vp, C.c_int]),
    "dmel_convnext_set_train_precision": (C.c_int, [vp, C.c_int]),
    "dmel_convnext_finalize": (C.c_int, [vp]),
    "dmel_convnext_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_convnext_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_convnext_train_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_convnext_grad_floats": (C.c_int64, [vp]),
    "dmel_convnext_grad_slot": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dmel_convnext_forward_train": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_convnext_backward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_wavenet_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dmel_wavenet_destroy": (None, [vp]),
    "dmel_wavenet_set_precision": (C.c_int, [vp, C.c_int]),
    "dmel_wavenet_enable_training": (C.c_int, [vp, C.c_int]),
    "dmel_wavenet_set_train_precision": (C.c_int, [vp, C.c_int]),
    "dmel_wavenet_refresh": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp), vp]),
    "dmel_wavenet_train_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_wavenet_grad_floats": (C.c_int64, [vp]),
    "dmel_wavenet_grad_slot": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dmel_wavenet_forward_train": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_wavenet_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_wavenet_backward_hooked": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp, GRAD_READY_FN, vp]),
    "dmel_wavenet_stream_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int64, i64p, i64p, vp]),
    "dmel_wavenet_set_tensor": (C.c_int, [vp, C.c_char_p, vp, i64p, C.c_int]),
    "dmel_wavenet_finalize": (C.c_int, [vp]),
    "dmel_wavenet_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_wavenet_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int64, vp, vp, C.c_int, vp, C.c_size_t, vp]),
    "dmel_quantizer_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]),
    "dmel_quantizer_destroy": (None, [vp]),
    "dmel_quantizer_set_strict": (C.c_int, [vp, C.c_int]),
    "dmel_quantizer_set_tensor": (C.c_int, [vp, C.c_char_p, vp, i64p, C.c_int]),
    "dmel_quantizer_finalize": (C.c_int, [vp]),
    "dmel_quantizer_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_quantizer_encode": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_quantizer_encode_ex": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_quantizer_decode": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_quantizer_enable_training": (C.c_int, [vp, C.c_int]),
    "dmel_quantizer_set_train_precision": (C.c_int, [vp, C.c_int]),
    "dmel_quantizer_refresh": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(vp), vp]),
    "dmel_quantizer_train_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_quantizer_grad_floats": (C.c_int64, [vp]),
    "dmel_quantizer_grad_slot": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dmel_quantizer_forward_train": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_quantizer_backward": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_mask_add_quality_f32": (C.c_int, [vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int64, vp]),
    "dmel_bigvgan_create": (C.c_int, [C.POINTER(vp), C.POINTER(BigVGANConfig)]),
    "dmel_bigvgan_destroy": (None, [vp]),
    "dmel_bigvgan_set_tensor": (C.c_int, [vp, C.c_char_p, vp, i64p, C.c_int]),
    "dmel_bigvgan_finalize": (C.c_int, [vp]),
    "dmel_bigvgan_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_bigvgan_set_streams": (C.c_int, [vp, C.c_int]),
    "dmel_stft_set_exclusive_cu": (C.c_int, [C.c_int]),
    "dmel_bigvgan_set_precision": (C.c_int, [vp, C.c_int]),
    "dmel_bigvgan_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_conv_create": (C.c_int, [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dmel_conv_destroy": (None, [vp]),
    "dmel_conv_set_precision": (C.c_int, [vp, C.c_int]),
    "dmel_conv_transpose1d_create": (C.c_int, [C.POINTER(vp), vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dmel_conv_transpose1d_destroy": (None, [vp]),
    "dmel_conv_transpose1d_set_precision": (C.c_int, [vp, C.c_int]),
    "dmel_conv_transpose1d_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_conv_post_f32": (C.c_int, [vp, vp, C.c_float, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]),
    "dmel_conv_backward_data": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_conv_backward_weight": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_conv_forward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, vp]),
    "dmel_stft_grad_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]),
    "dmel_stft_grad_destroy": (None, [vp]),
    "dmel_stft_grad_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int64]),
    "dmel_stft_magnitude_backward_f32": (C.c_int, [vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_int, C.c_int64, vp, C.c_size_t, vp]),
    "dmel_collate_peak_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int64, C.c_float, vp]),
    "dmel_conv_snake_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp]),
    "dmel_prof_enable": (C.c_int, [C.c_int]),
    "dmel_prof_reset": (C.c_int, []),
    "dmel_prof_read": (C.c_int, [C.c_char_p, i64p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dmel_prof_read_ex": (C.c_int, [C.c_char_p, i64p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m dmel_codec_amd.build). "
                "dmel_codec_amd has no CPU or eager fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dmel_last_error().decode(errors="replace")
        raise RuntimeError(f"libdmel_hip {what} failed (code {rc}): {msg}")


def require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (got {t.device}); dmel_codec_amd has no CPU path")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def shape_array(shape: Sequence[int]):
    return (C.c_int64 * max(1, len(shape)))(*shape)


def set_tensors(set_fn, handle, state: dict, what: str) -> None:
    """Hand every fp32 tensor of a state dict to a native handle under its key name."""
    for key, val in state.items():
        t = val.detach().to(device="cpu", dtype=torch.float32).contiguous()
        check(set_fn(handle, key.encode(), t.data_ptr(), shape_array(t.shape), t.ndim), f"{what}.set_tensor({key})")


class Workspace:
    """Grow-only device scratch owned by the Python side (the library never allocates activations)."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != torch.device(device):
            self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        return self.buf


def prof_enable(on: bool) -> None:
    check(lib().dmel_prof_enable(int(on)), "prof_enable")


def prof_reset() -> None:
    check(lib().dmel_prof_reset(), "prof_reset")


def prof_read(family: str):
    n = C.c_int64()
    ms, fl, by, iss = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    check(lib().dmel_prof_read_ex(family.encode(), C.byref(n), C.byref(ms), C.byref(fl), C.byref(by), C.byref(iss)), "prof_read")
    return {"launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value, "issue_flops": iss.value}
