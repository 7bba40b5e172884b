"""Experiment: decode the batch as two halves on two streams (two codec instances) vs one full-batch decode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
c1 = bench.build("cfg2").to(dev)
c2 = bench.build("cfg2").to(dev)
audio = bench.synth_audio(32, 24000, 1234).to(dev)
lens = torch.full((32,), 24000, device=dev, dtype=torch.int64)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def full():
    ids, il = c1.encode(audio, lens)
    return c1.decode(ids, il, return_audios=True)[0]

def split():
    ids, il = c1.encode(audio, lens)
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = c1.decode(ids[:16], il[:16], return_audios=True)[0]
    with torch.cuda.stream(s2):
        b = c2.decode(ids[16:], il[16:], return_audios=True)[0]
    cur.wait_stream(s1); cur.wait_stream(s2)
    return a, b

for fn, name in ((full, "full"), (split, "split"), (full, "full"), (split, "split")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(name, "ms/step", round((time.perf_counter() - t0) * 100, 2))
