"""Mel-image discriminator: parameter container.  Mirrors dmel_codec/models/modules/discriminator.py:6-35
(reference): six weight-normed Conv2d with SiLU in between, state-dict keys
`blocks.{i}.bias`, `blocks.{i}.parametrizations.weight.original0|1`.  It is used only by training_step
(codec_lit_modules.py:214-215), which is SURVEY.md 8f rank 1 and not built yet: forward() raises."""
from __future__ import annotations

from torch import nn
from torch.nn.utils.parametrizations import weight_norm


class Discriminator(nn.Module):
    def __init__(self):
        super().__init__()
        convs = [(1, 64, (3, 9), 1, (1, 4)), (64, 128, (3, 9), (1, 2), (1, 4)), (128, 256, (3, 9), (1, 2), (1, 4)),
                 (256, 512, (3, 9), (1, 2), (1, 4)), (512, 1024, (3, 3), 1, (1, 1)), (1024, 1, (3, 3), 1, (1, 1))]
        blocks = []
        for idx, (cin, cout, k, stride, pad) in enumerate(convs):
            blocks.append(weight_norm(nn.Conv2d(cin, cout, k, stride, pad)))
            if idx != len(convs) - 1:
                blocks.append(nn.SiLU(inplace=True))
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        raise NotImplementedError("the discriminator only runs inside training_step (SURVEY.md 8f rank 1), not built yet")
