import sys, os, copy, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from test_gpu_parity import make_codec
dev = torch.device("cuda:0")
if os.environ.get("DMEL_DBG_EXCL") == "1":
    from dmel_codec_amd import _lib as _l
    _l.check(_l.lib().dmel_stft_set_exclusive_cu(1), "excl")
    print("exclusive CU mode on", flush=True)
codec = make_codec(720, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3).to(dev)
codec2 = copy.deepcopy(codec)
gen = torch.Generator().manual_seed(5)
L = 24000
batches = [(0.3 * torch.randn(3, 1, L, generator=gen)).to(dev) for _ in range(4)]
lens = [torch.tensor([L, L - 700 * (i + 1), L // 2], device=dev) for i in range(4)]
cs = [codec, codec2]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
MODE = sys.argv[1] if len(sys.argv) > 1 else "full"

def stages(c, a, l):
    mel = c.encode_mel_transform(a)
    if MODE == "stft":
        return (mel,)
    feats, ml = c.encode_unquantized(a, l)
    if MODE == "feat":
        return mel, feats
    ids, il = c.encode(a, l)
    return mel, feats, ids

ref = [stages(cs[i % 2], batches[i], lens[i]) for i in range(4)]
torch.cuda.synchronize()
names = ["mel", "features", "ids"]
nbad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    outs = []
    for i in range(4):
        k = i % 2
        streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[k]):
            outs.append(stages(cs[k], batches[i], lens[i]))
    torch.cuda.synchronize()
    for i in range(4):
        for j in range(len(ref[i])):
            if not torch.equal(outs[i][j], ref[i][j]):
                nbad += 1
                d = (outs[i][j] != ref[i][j]).nonzero()
                cols = sorted(set(d[:, -1].tolist())); rows = sorted(set(d[:, 1].tolist())); items = sorted(set(d[:, 0].tolist()))
                print(f"trial {trial} batch {i} lane {i % 2} {names[j]} shape {tuple(ref[i][j].shape)}: {d.shape[0]} values differ; items {items}; "
                      f"rows {rows[:8]}{'...' if len(rows) > 8 else ''} ({len(rows)}); frames {cols[:12]} ({len(cols)}); "
                      f"max diff {float((outs[i][j].float() - ref[i][j].float()).abs().max()):.4f}", flush=True)
print("mode", MODE, "mismatches", nbad)
