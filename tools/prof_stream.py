"""One 120 s stream at 64-token chunks, batch 1, sequential pushes: run under `rocprofv3 --kernel-trace --stats` to see where a push's time goes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
codec = bench.build("cfg2r").to(dev)
ids = torch.randint(0, 175, (1, 10, 2813), generator=torch.Generator().manual_seed(6), dtype=torch.int32).to(dev)
for rep in range(2):
    n = 0
    for a, m in codec.decode_stream(ids, None, chunk_tokens=64):
        n += a.shape[-1]
    torch.cuda.synchronize()
print("pushes per pass", (2813 + 63) // 64, "samples", n)
