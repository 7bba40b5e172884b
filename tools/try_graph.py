"""Experiment: capture one encode+decode step in a HIP graph (torch.cuda.CUDAGraph) and compare replay with eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
codec = bench.build("cfg2").to(dev)
audio = bench.synth_audio(32, 24000, 1234).to(dev)
lens = torch.full((32,), 24000, device=dev, dtype=torch.int64)

def step():
    ids, il = codec.encode(audio, lens)
    wav, _ = codec.decode(ids, il, return_audios=True)
    return ids, wav

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) * 100)

g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    ids_g, wav_g = step()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print("graph ms/step", (time.perf_counter() - t0) * 100)
ids_e, wav_e = step()
print("ids equal", torch.equal(ids_e, ids_g))
