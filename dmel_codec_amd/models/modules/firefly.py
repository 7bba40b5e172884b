"""ConvNeXtBlock / LayerNorm parameter containers.  Mirrors dmel_codec/models/modules/firefly.py:306-402 (reference);
only the two classes the codec path uses are provided -- the arithmetic runs inside the quantiser's native calls
(csrc/small_ops.hip dwconv_ln + two implicit-GEMM launches)."""
from __future__ import annotations

import torch
from torch import nn


class LayerNorm(nn.Module):
    """firefly.py:306-333 (channels_last form; parameter container)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format != "channels_last":
            raise NotImplementedError("only the channels_last form is on the codec path")
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)


class ConvNeXtBlock(nn.Module):
    """firefly.py:337-402 (parameter container)."""

    def __init__(self, dim: int, drop_path: float = 0.0, layer_scale_init_value: float = 1e-6, mlp_ratio: float = 4.0,
                 kernel_size: int = 7, dilation: int = 1):
        super().__init__()
        if kernel_size != 7 or dilation != 1 or mlp_ratio != 4.0 or drop_path != 0.0 or layer_scale_init_value <= 0:
            raise NotImplementedError("only the default ConvNeXtBlock (k=7, mlp 4x, layer scale) is on the codec path")
        self.dwconv = nn.Conv1d(dim, dim, kernel_size=kernel_size, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, int(mlp_ratio * dim))
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(int(mlp_ratio * dim), dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True)
