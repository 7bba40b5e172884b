"""Several batches in flight on one GPU: `CodecLanes`.

The codec's kernels are of two kinds: the vocoder's, whose grids fill the chip, and the WaveNets' (20 + 20 dependent layers over ~90
frames per item), whose 160-290 workgroups leave most of the SIMDs with one wave and long dependent K loops.  Run back to back, the second
kind leaves the GPU mostly idle for a fifth of a step.  Independent BATCHES do not depend on each other, so the encode + decoder-WaveNet
phase of batch i + 1 can run under the vocoder phase of batch i: a lane is a replica of the codec (own native handles, own workspaces --
a few hundred MB of the 288 GB) with its own HIP stream; batches are dealt to the lanes round-robin and the hardware interleaves the
lanes' kernels.  Outputs are those of the single codec, bit for bit (tests/test_gpu_parity.py::test_lanes_match_sequential): the lanes run
the same kernels on the same weights, and host-side random draws (the decoder's Gaussian input) happen in submission order.

The reference has no counterpart (it decodes one batch at a time, `codec_lit_modules.py:462-531`); this is the serving-side form of
SURVEY 8(e) "independent utterances, no collective", applied inside one GPU."""
from __future__ import annotations

import copy
from typing import Callable, List, Optional

import torch


class LaneResult:
    """What a lane hands back: the tensors of one batch and the event after which they are valid."""

    def __init__(self, values, event: torch.cuda.Event, stream: torch.cuda.Stream):
        self.values, self.event, self.stream = values, event, stream

    def wait(self):
        """Make the CURRENT stream wait for the batch and return its tensors (no host synchronisation)."""
        cur = torch.cuda.current_stream()
        cur.wait_event(self.event)
        for v in _tensors(self.values):
            v.record_stream(cur)          # allocated on the lane's stream, consumed on this one
        return self.values

    def synchronize(self):
        self.event.synchronize()
        return self.values


def _tensors(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (tuple, list)):
        for v in x:
            yield from _tensors(v)
    elif isinstance(x, dict):
        for v in x.values():
            yield from _tensors(v)


class CodecLanes:
    """`lanes = CodecLanes(codec, 2); r = lanes.roundtrip(audio, lengths); ...; ids, wav = r.wait()`.

    codec: a VQGAN on a CUDA device (lane 0 IS this object; the other lanes are deep copies made here: same weights, frozen).  Weights
    changed afterwards (load_state_dict, an optimiser step) must be followed by `refresh()`."""

    def __init__(self, codec, n_lanes: int = 2):
        if n_lanes < 1:
            raise ValueError("n_lanes must be >= 1")
        dev = next(codec.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("CodecLanes needs the codec on a CUDA device (there is no CPU path)")
        self.device = dev
        if n_lanes > 1:
            # kernels of different lanes share CUs; the STFT kernel must not share one with the convolution kernels (include/dmel_hip.h,
            # dmel_stft_set_exclusive_cu: a measured corruption of single frames, cured by giving its workgroups a CU's whole LDS)
            from . import _lib
            _lib.check(_lib.lib().dmel_stft_set_exclusive_cu(1), "stft_set_exclusive_cu")
        self.codecs = [codec] + [copy.deepcopy(codec).eval() for _ in range(n_lanes - 1)]
        with torch.cuda.device(dev):
            self.streams: List[torch.cuda.Stream] = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
        self._next = 0

    def __len__(self) -> int:
        return len(self.codecs)

    def configure(self, fn: Callable) -> None:
        """Apply `fn(codec)` to every lane (precision switches, vocoder stream count, ...)."""
        for c in self.codecs:
            fn(c)

    def refresh(self) -> None:
        """Copy lane 0's weights into the other lanes (after load_state_dict / training)."""
        sd = self.codecs[0].state_dict()
        for c in self.codecs[1:]:
            c.load_state_dict(sd, strict=True)

    def submit(self, fn: Callable, *args, lane: Optional[int] = None) -> LaneResult:
        """Run `fn(codec_of_the_lane, *args)` on the next lane's stream.  Tensor arguments must be valid on the current stream."""
        k = self._next if lane is None else lane
        if lane is None:
            self._next = (self._next + 1) % len(self.codecs)
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))      # inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            out = fn(self.codecs[k], *args)
            ev = torch.cuda.Event()
            ev.record(s)
        for a in _tensors(args):
            a.record_stream(s)
        return LaneResult(out, ev, s)

    def encode(self, audios, audio_lengths) -> LaneResult:
        return self.submit(lambda c, a, l: c.encode(a, l), audios, audio_lengths)

    def decode(self, indices, feature_lengths, return_audios: bool = True) -> LaneResult:
        return self.submit(lambda c, i, l: c.decode(i, l, return_audios=return_audios), indices, feature_lengths)

    def roundtrip(self, audios, audio_lengths) -> LaneResult:
        """encode -> ids -> decode to audio of one batch on one lane: (ids, feature_lengths, waveform)."""
        def fn(c, a, l):
            ids, il = c.encode(a, l)
            wav, _ = c.decode(ids, il, return_audios=True)
            return ids, il, wav
        return self.submit(fn, audios, audio_lengths)

    def synchronize(self) -> None:
        for s in self.streams:
            s.synchronize()
