"""`python train_codec.py ...` from the repository root: the reference's dmel_codec/train_codec.py entry point on the MI355X
implementation (dmel_codec_amd/train_codec.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from dmel_codec_amd.train_codec import cli  # noqa: E402

if __name__ == "__main__":
    cli(sys.argv[1:])
