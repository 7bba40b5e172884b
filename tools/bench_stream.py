"""Streaming decode benchmark (BASELINE.json configs[4], codec side): token ids (1, G, T4) arrive in chunks, audio leaves as soon as its
right context exists (VQGAN.decode_stream: WaveNet state carry + windowed vocoder).  Reports the latency from the first token to the
first audio, the sustained audio-seconds per second at batch 1, and the windowed (stateless, halo re-run) form for comparison."""
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
ONLY_PIPE = "--pipeline-only" in sys.argv


def pipeline_section():
    # decode_stream(pipeline=True): the vocoder of chunk i on its own stream, overlapping the decoder WaveNet of chunk i + 1 (pieces are
    # yielded one chunk later); 120 s stream, batch 1 and 16, against the sequential generator
    T4L = 2813
    gl = torch.Generator().manual_seed(6)
    ids_long = torch.randint(0, 175, (1, 10, T4L), generator=gl, dtype=torch.int32).to(dev)
    for B in ((1,) if "--batch1" in sys.argv else (1, 16)):
        ids_b = ids_long.expand(B, -1, -1).contiguous()
        for chunk in (32, 64, 128):
            for pipe in (False, True):
                for rep in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    n = 0
                    for a, m in codec.decode_stream(ids_b, None, chunk_tokens=chunk, pipeline=pipe):
                        n += a.shape[-1]
                    torch.cuda.synchronize()
                    el = time.perf_counter() - t0
                print(json.dumps({"mode": "decode_stream pipeline" if pipe else "decode_stream", "batch": B, "chunk_tokens": chunk,
                                  "audio_s_per_stream": round(n / 24000, 2), "ms_per_chunk": round(el * 1e3 / ((T4L + chunk - 1) // chunk), 3),
                                  "audio_sec_per_sec_all_streams": round(B * n / 24000 / el, 1)}), flush=True)


if ONLY_PIPE:
    pipeline_section()
    sys.exit(0)
for chunk in (8, 32, 64, 128):
    list(codec.decode_stream(ids, flen, chunk_tokens=chunk))      # warm-up (handles, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first, first_tokens = None, None
    n = fed = 0
    dec = codec.streaming_decoder(1, flen)
    for a0 in range(0, T4, chunk):
        a, m = dec.push(ids[:, :, a0:a0 + chunk])
        fed = min(T4, a0 + chunk)
        if first is None and a.shape[-1]:
            a.cpu()                               # first audio: include the device->host hand-off
            first, first_tokens = time.perf_counter() - t0, fed
        n += a.shape[-1]
    a, m = dec.finish()
    n += a.shape[-1]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"mode": "state carry", "chunk_tokens": chunk, "first_audio_ms": round(first * 1e3, 2), "first_audio_after_tokens": first_tokens,
                      "audio_s": round(n / 24000, 2), "audio_sec_per_sec": round(n / 24000 / el, 1)}), flush=True)
# the same stream with steady-state pushes replayed as ONE HIP graph each (StreamingDecoder(graph_chunk_tokens=...)), batch 1 and a batch of
# independent streams (what "replicas" of BASELINE config 5 share one GPU as).  A longer stream (120 s) so that the one-time capture and the
# start-up pushes do not dominate; the steady-state time per push is timed separately over the last pushes.
T4L = 2813
gl = torch.Generator().manual_seed(6)
ids_long = torch.randint(0, 175, (1, 10, T4L), generator=gl, dtype=torch.int32).to(dev)
for B in (1, 16):
    ids_b = ids_long.expand(B, -1, -1).contiguous()
    for chunk in (32, 64, 128):
        for graph in (False, True):
            for rep in range(2):                # first pass warms handles / workspaces
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dec = codec.streaming_decoder(B, None, True, graph_chunk_tokens=chunk if graph else None)
                n = 0
                starts = list(range(0, T4L, chunk))
                t_steady = None
                for i, a0 in enumerate(starts):
                    if i == len(starts) - 21:
                        torch.cuda.synchronize()
                        t_steady = time.perf_counter()
                    a, m = dec.push(ids_b[:, :, a0:a0 + chunk])
                    n += a.shape[-1]
                    if i == len(starts) - 2:
                        torch.cuda.synchronize()
                        steady_ms = (time.perf_counter() - t_steady) * 1e3 / 20
                a, m = dec.finish()
                n += a.shape[-1]
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            print(json.dumps({"mode": "state carry + graph replay" if graph else "state carry", "batch": B, "chunk_tokens": chunk,
                              "graph_replays": dec.graph_replays, "pushes": len(starts), "steady_ms_per_push": round(steady_ms, 3),
                              "steady_audio_sec_per_sec_all_streams": round(B * chunk * 4 * 256 / 24000 / (steady_ms * 1e-3), 1),
                              "audio_s_per_stream": round(n / 24000, 2), "audio_sec_per_sec_all_streams": round(B * n / 24000 / el, 1)}), flush=True)
for chunk in (32, 64, 128):
    codec.decode_chunked(ids, flen, chunk_tokens=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, _ = codec.decode_chunked(ids, flen, chunk_tokens=chunk)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"mode": "windows re-run with halo", "chunk_tokens": chunk, "halo_tokens": codec.STREAM_HALO_TOKENS,
                      "audio_sec_per_sec": round(a.shape[-1] / 24000 / el, 1)}), flush=True)
codec.decode(ids, flen, return_audios=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
a, _ = codec.decode(ids, flen, return_audios=True)
torch.cuda.synchronize()
print(json.dumps({"whole_sequence_decode_ms": round((time.perf_counter() - t0) * 1e3, 2), "audio_s": round(a.shape[-1] / 24000, 2)}))
pipeline_section()
