"""Snake / SnakeBeta parameter containers.  Mirrors dmel_codec/models/modules/bigvgan/activations.py:9-126
(reference).  The arithmetic x + 1/(b+1e-9) sin^2(a x) runs inside the fused anti-alias kernel (csrc/aa_snake.hip)."""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import Parameter


class Snake(nn.Module):
    """activations.py:9-62"""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        self.alpha = Parameter((torch.zeros if alpha_logscale else torch.ones)(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001


class SnakeBeta(nn.Module):
    """activations.py:65-126"""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        make = torch.zeros if alpha_logscale else torch.ones
        self.alpha = Parameter(make(in_features) * alpha)
        self.beta = Parameter(make(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001
