"""dmel_codec/train_codec.py of the reference, served by dmel_codec_amd.train_codec (python -m dmel_codec.train_codec)."""
import sys

from dmel_codec_amd.train_codec import cli, get_config, main  # noqa: F401

if __name__ == "__main__":
    cli(sys.argv[1:])
