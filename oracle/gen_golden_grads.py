"""Gradient fixtures for the training path, from autograd through the REFERENCE's own modules (build container only; the reference
does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_grads.py

tests/golden/train_grads_*.npz hold data only: seeded inputs, weights (reference key names), the upstream gradient dy and the
gradients the reference module's autograd produced for the inputs and every parameter.  They pin the oracle's autograd (which the GPU
tests use as the training truth) to the reference: tests/test_oracle_golden.py::test_training_gradients_match_reference.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle.gen_golden import randomise, save  # noqa: E402


def grads_of(m, y, dy, inputs):
    m.zero_grad()
    for t in inputs.values():
        t.grad = None
    (y * dy).sum().backward()
    outs = {"y": y.detach()}
    for k, t in inputs.items():
        outs["d_" + k] = t.grad.detach().clone()
    for k, p in m.named_parameters():
        if p.grad is not None:
            outs["g/" + k] = p.grad.detach().clone()
    return outs


def main() -> None:
    torch.set_num_threads(1)
    from dmel_codec.models.modules.wavenet import WaveNet
    from dmel_codec.models.modules.firefly import ConvNeXtBlock
    from dmel_codec.models.modules.bigvgan import activations
    from dmel_codec.models.modules.bigvgan.alias_free_activation.torch.act import Activation1d

    g = torch.Generator().manual_seed(20251004)

    m = WaveNet(input_channels=24, output_channels=12, residual_channels=24, residual_layers=4, dilation_cycle=4,
                condition_channels=24)
    randomise(m, g)
    x = torch.randn(2, 24, 31, generator=g, requires_grad=True)
    c = torch.randn(2, 24, 31, generator=g, requires_grad=True)
    dy = torch.randn(2, 12, 31, generator=g)
    outs = grads_of(m, m(x, condition=c), dy, {"x": x, "cond": c})
    save("train_grads_wavenet", {"pinned": True, "ref": "autograd through models/modules/wavenet.py:204-225", "n_layers": 4,
                                 "dilation_cycle": 4}, m.state_dict(), {"x": x, "cond": c, "dy": dy}, outs)

    m = ConvNeXtBlock(dim=20)
    randomise(m, g)
    x = torch.randn(3, 20, 27, generator=g, requires_grad=True)
    dy = torch.randn(3, 20, 27, generator=g)
    outs = grads_of(m, m(x), dy, {"x": x})
    save("train_grads_convnext", {"pinned": True, "ref": "autograd through models/modules/firefly.py:383-402"}, m.state_dict(),
         {"x": x, "dy": dy}, outs)

    m = Activation1d(activation=activations.SnakeBeta(5, alpha_logscale=True))
    randomise(m, g)
    x = (torch.randn(2, 5, 77, generator=g) * 1.5).requires_grad_()
    dy = torch.randn(2, 5, 77, generator=g)
    outs = grads_of(m, m(x), dy, {"x": x})
    save("train_grads_activation1d", {"pinned": True, "ref": "autograd through bigvgan/alias_free_activation/torch/act.py:25-30"},
         m.state_dict(), {"x": x, "dy": dy}, outs)


def disc_fixture() -> None:
    """Discriminator: forward logits and autograd gradients of the reference class on seeded weights (the fixture carries the seed, the
    input, the logits, d input, the gradients of the small tensors and seeded random projections of the large ones)."""
    from dmel_codec.models.modules.discriminator import Discriminator
    from oracle import ref_cpu
    seed = 4242
    m = Discriminator()
    m.load_state_dict(ref_cpu.seeded_discriminator_sd(seed))
    g = torch.Generator().manual_seed(99)
    x = torch.randn(2, 80, 37, generator=g, requires_grad=True)
    y = m(x)
    dy = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward()
    outs = {"y": y.detach(), "d_x": x.grad.detach()}
    for k, p in m.named_parameters():
        if p.numel() <= 4096:
            outs["g/" + k] = p.grad.detach()
        else:
            r = torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel()))
            outs["proj/" + k] = (p.grad.detach() * r).sum().reshape(1)
    save("train_grads_discriminator", {"pinned": True, "ref": "models/modules/discriminator.py:6-35 (+ autograd)", "seed": seed}, {},
         {"x": x, "dy": dy}, outs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "disc":
        disc_fixture()
        sys.exit(0)
    main()
