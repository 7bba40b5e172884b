"""Experiment: two batches in flight on one GPU (two codec replicas with their own workspaces, each on its own stream) against one.
    python tools/try_lanes.py [--lanes 2] [--steps 20]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--streams", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
L = 24000
for lanes in sorted({1, args.lanes}):
    codecs = [bench.build("cfg2").to(dev) for _ in range(lanes)]
    for c in codecs:
        c.set_decode_precision("fp32")
        c.vocoder.set_streams(args.streams)
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    audio = [bench.synth_audio(32, L, 1234 + i).to(dev) for i in range(lanes)]
    lens = torch.full((32,), L, device=dev, dtype=torch.int64)

    def step(i):
        k = i % lanes
        with torch.cuda.stream(streams[k]):
            ids, il = codecs[k].encode(audio[k], lens)
            wav, _ = codecs[k].decode(ids, il, return_audios=True)
        return wav

    for i in range(2 * lanes + 2):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"lanes={lanes} streams/vocoder={args.streams}: {dt / args.steps * 1e3:.3f} ms per step, {32 * args.steps / dt:.1f} audio-s/s (host enqueue {t_host / args.steps * 1e3:.2f} ms per step)", flush=True)
    del codecs
