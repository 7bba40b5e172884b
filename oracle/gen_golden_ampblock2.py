"""Golden fixtures for the AMPBlock2 form of BigVGAN (`resblock: "2"`, bigvgan.py:150-241), generated from the REFERENCE's own classes.

Run only in the build container (needs /root/reference; it does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ampblock2.py

A separate script (own seed) so that the fixtures of oracle/gen_golden.py keep their bytes."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle.gen_golden import randomise, save  # noqa: E402


def main() -> None:
    from dmel_codec.models.modules.bigvgan.bigvgan import AMPBlock2, BigVGAN
    from dmel_codec.models.modules.bigvgan.env import AttrDict
    g = torch.Generator().manual_seed(20262)
    h = AttrDict({"snake_logscale": True, "use_cuda_kernel": False})
    m = AMPBlock2(h, 16, 7, (1, 3, 5), activation="snakebeta").eval()
    randomise(m, g)
    x = torch.randn(2, 16, 50, generator=g)
    with torch.no_grad():
        y = m(x)
    save("ampblock2", {"pinned": True, "ref": "models/modules/bigvgan/bigvgan.py:232-237", "k": 7, "dilations": [1, 3, 5]},
         m.state_dict(), {"x": x}, {"y": y})
    hd = {"num_mels": 20, "upsample_rates": [4, 2], "upsample_kernel_sizes": [8, 4], "upsample_initial_channel": 32, "resblock": "2",
          "resblock_kernel_sizes": [3, 7, 11], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "activation": "snakebeta",
          "snake_logscale": True, "use_tanh_at_final": True, "use_bias_at_final": True}
    m = BigVGAN(AttrDict(dict(hd))).eval()
    randomise(m, g)
    x = torch.randn(2, 20, 9, generator=g)
    with torch.no_grad():
        y = m(x)
    save("bigvgan_tiny_ampblock2", {"pinned": True, "ref": "models/modules/bigvgan/bigvgan.py:150-241,367-393", "h": hd},
         m.state_dict(), {"mel": x}, {"audio": y})


if __name__ == "__main__":
    main()
