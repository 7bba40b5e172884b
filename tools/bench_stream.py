"""Streaming decode benchmark (BASELINE.json configs[4], codec side): token ids (1, G, T4) -> audio in halo'd chunks.
Reports the latency to the first audio chunk and the sustained audio-seconds per second at batch 1."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
codec = bench.build("cfg2r").to(dev)          # LM configs use the 100-mel / 10-group codec (config/lm/lm_config.yaml)
g = torch.Generator().manual_seed(5)
T4 = 469                                      # 20 s of audio at 23.4 token frames per second
ids = torch.randint(0, 175, (1, 10, T4), generator=g, dtype=torch.int32).to(dev)
flen = torch.tensor([T4], device=dev)
for chunk in (32, 64, 128):
    list(codec.decode_stream(ids, flen, chunk_tokens=chunk))      # warm-up (handles, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = None
    n = 0
    for a, m in codec.decode_stream(ids, flen, chunk_tokens=chunk):
        a.cpu() if first is None else None       # first chunk: include the device->host hand-off
        if first is None:
            first = time.perf_counter() - t0
        n += a.shape[-1]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"chunk_tokens": chunk, "halo_tokens": codec.STREAM_HALO_TOKENS, "first_chunk_ms": round(first * 1e3, 2),
                      "audio_s": round(n / 24000, 2), "audio_sec_per_sec": round(n / 24000 / el, 1)}))
t0 = time.perf_counter()
a, _ = codec.decode(ids, flen, return_audios=True)
torch.cuda.synchronize()
print(json.dumps({"whole_sequence_decode_ms": round((time.perf_counter() - t0) * 1e3, 2), "audio_s": round(a.shape[-1] / 24000, 2)}))
