"""AttrDict.  Mirrors dmel_codec/models/modules/bigvgan/env.py:8-11 (reference)."""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self

    # copies and pickles go through the constructor: the default protocol restores the items but not `__dict__ = self`, and attribute
    # access on the copy fails (copy.deepcopy(BigVGAN) -- dmel_codec_amd.pipeline.CodecLanes -- needs it)
    def __reduce__(self):
        return (AttrDict, (dict(self),))

    def __deepcopy__(self, memo):
        import copy
        return AttrDict(copy.deepcopy(dict(self), memo))
