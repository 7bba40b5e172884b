"""Mirrors the two helpers of dmel_codec/models/modules/bigvgan/utils.py the generator uses (:45-48, :57-58)."""


def init_weights(m, mean=0.0, std=0.01):
    if m.__class__.__name__.find("Conv") != -1:
        m.weight.data.normal_(mean, std)


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)
