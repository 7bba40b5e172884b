"""Debug aid: every native workspace gets a 1 MiB guard region behind the bytes the library asked for (pattern 0x5A); after encode + decode the
guards must be intact.  A write behind a workspace lands in whatever tensor the caching allocator placed there -- with several batches in
flight (CodecLanes) that is another batch's live tensor."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from dmel_codec_amd import _lib
GUARD = 1 << 20
spaces = []

def get(self, nbytes, device):
    nbytes = max(int(nbytes), 256)
    if getattr(self, "full", None) is None or self.req < nbytes or self.full.device != torch.device(device):
        self.full = torch.full((nbytes + GUARD,), 0x5A, dtype=torch.uint8, device=device)
        self.req = nbytes
        spaces.append(self)
    self.buf = self.full[:self.req]
    return self.buf

_lib.Workspace.get = get

def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for w in spaces:
        g = w.full[w.req:]
        nz = (g != 0x5A).nonzero()
        if nz.numel():
            bad += 1
            print(f"[{tag}] workspace of {w.req} bytes: {nz.shape[0]} guard bytes overwritten, first at +{int(nz[0])}, last at +{int(nz[-1])}", flush=True)
            g.fill_(0x5A)
    print(f"[{tag}] {len(spaces)} workspaces, {bad} violated", flush=True)

dev = torch.device("cuda:0")
from test_gpu_parity import make_codec
codec = make_codec(720, n_mels=80, dmel_groups=8, encoder_layers=2, decoder_layers=3).to(dev)
g = torch.Generator().manual_seed(5)
a = (0.3 * torch.randn(3, 1, 24000, generator=g)).to(dev)
l = torch.tensor([24000, 23300, 12000], device=dev)
ids, il = codec.encode(a, l); check("small encode")
wav, _ = codec.decode(ids, il, return_audios=True); check("small decode")
import bench
big = bench.build("cfg2").to(dev)
audio = bench.synth_audio(32, 24000, 1).to(dev)
lens = torch.full((32,), 24000, device=dev, dtype=torch.int64)
ids, il = big.encode(audio, lens); check("cfg2 encode")
wav, _ = big.decode(ids, il, return_audios=True); check("cfg2 decode")
big.vocoder.set_streams(1)
wav, _ = big.decode(ids, il, return_audios=True); check("cfg2 decode, one stream")
dec = big.streaming_decoder(1, None, True)
for a0 in range(0, 256, 64):
    dec.push(ids[:1, :, a0 % 20:a0 % 20 + 3].repeat(1, 1, 22)[:, :, :64].contiguous())
check("cfg2 streaming pushes")
