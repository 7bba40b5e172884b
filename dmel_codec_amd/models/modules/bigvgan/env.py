"""AttrDict.  Mirrors dmel_codec/models/modules/bigvgan/env.py:8-11 (reference)."""


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self
