"""Soak test of dmel_codec_amd.pipeline.CodecLanes on the bench workload (cfg2, batch 32 x 1 s, three lanes): N rounds of eight different batches,
every roundtrip (ids AND waveform, decoder noise injected so that it is reproducible) compared bit for bit with the same batch through the single
codec.  Prints the number of mismatching roundtrips (must be 0).    python tools/soak_lanes.py [--rounds 30] [--lanes 3]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmel_codec_amd.pipeline import CodecLanes

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--lanes", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
codec = bench.build("cfg2").to(dev)
codec.vocoder.set_streams(1)
L = 24000
NB = 8
audio = [bench.synth_audio(32, L, 100 + i).to(dev) for i in range(NB)]
lens = torch.full((32,), L, device=dev, dtype=torch.int64)
gen = torch.Generator(device=dev).manual_seed(7)
noise = [torch.randn(32, 560, 92, device=dev, generator=gen) for _ in range(NB)]

def rt(c, a, l, nz):
    ids, il = c.encode(a, l)
    wav, _ = c.decode(ids, il, return_audios=True, noise=nz)
    return ids, wav

ref = [tuple(t.clone() for t in rt(codec, audio[i], lens, noise[i])) for i in range(NB)]
torch.cuda.synchronize()
lanes = CodecLanes(codec, args.lanes)
lanes.configure(lambda c: c.vocoder.set_streams(1))
bad_ids = bad_wav = n = 0
for r in range(args.rounds):
    res = [lanes.submit(rt, audio[i], lens, noise[i]) for i in range(NB)]
    for i, x in enumerate(res):
        ids, wav = x.wait()
        torch.cuda.synchronize()
        n += 1
        bad_ids += 0 if torch.equal(ids, ref[i][0]) else 1
        bad_wav += 0 if torch.equal(wav, ref[i][1]) else 1
print(f"{n} roundtrips through {args.lanes} lanes: {bad_ids} with different ids, {bad_wav} with a different waveform", flush=True)
sys.exit(1 if bad_ids or bad_wav else 0)
