"""The host packer of the fp16-split weight image (csrc/common.h: f32_to_f16_bits / f16_bits_to_f32; csrc/conv.h: pack_conv) must round
exactly like the device (v_cvt_f16_f32: nearest even, subnormals kept, overflow to inf), or the device-side re-pack after an optimiser step
would produce a different image.  numpy's float16 conversion is IEEE round-to-nearest-even: the reference here.  Also pins the property the
arithmetic rests on: hi + 2^-11 lo reproduces an fp32 number to 2^-22 relative (22 significant bits, as in the "3xTF32" construction; fp32 has 24)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "%s/dmel_codec_amd/csrc/common.h"
int main(int argc, char** argv) {
  FILE* in = fopen(argv[1], "rb"); FILE* out = fopen(argv[2], "wb");
  float x;
  while (fread(&x, 4, 1, in) == 1) {
    const uint16_t h = dmel::f32_to_f16_bits(x);
    const float b = dmel::f16_bits_to_f32(h);
    const uint16_t lo = dmel::f32_to_f16_bits((x - b) * dmel::kF16LoScale);
    fwrite(&h, 2, 1, out); fwrite(&b, 4, 1, out); fwrite(&lo, 2, 1, out);
  }
  fclose(in); fclose(out);
  return 0;
}
'''


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_host_fp16_split_matches_ieee_and_carries_22_bits(tmp_path):
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2 ** 32, size=400000, dtype=np.uint64).astype(np.uint32)
    wide = bits.view(np.float32)
    audio = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 12, 200000))).astype(np.float32)     # 1e-9 .. 1e5
    edge = np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 5.96e-8,
                     6.1e-5, np.inf, -np.inf], dtype=np.float32)
    x = np.concatenate([wide[np.isfinite(wide)], audio, edge])
    (tmp_path / "t.cpp").write_text(SRC % ROOT)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(tmp_path / "t.cpp"), "-o", str(exe)],
                   check=True, capture_output=True)
    x.tofile(tmp_path / "in.bin")
    subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], check=True)
    rec = np.fromfile(tmp_path / "out.bin", dtype=np.dtype([("h", "<u2"), ("b", "<f4"), ("lo", "<u2")]))
    assert len(rec) == len(x)
    with np.errstate(over="ignore"):
        ref_h = x.astype(np.float16)
    assert np.array_equal(rec["h"], ref_h.view(np.uint16))                      # the rounding of the first piece
    back = ref_h.astype(np.float32)
    assert np.array_equal(rec["b"].view(np.uint32), back.view(np.uint32))       # and its widening
    ok = np.isfinite(back)
    with np.errstate(over="ignore", invalid="ignore"):
        ref_lo = ((x - back) * np.float32(2048.0)).astype(np.float16)
    assert np.array_equal(rec["lo"][ok], ref_lo.view(np.uint16)[ok])
    # two pieces: |x - (hi + lo / 2048)| <= 2^-22 |x| wherever the first piece is a normal fp16 number (rms well below: ~2^-24)
    normal = ok & (np.abs(x) >= 2.0 ** -14) & (np.abs(x) < 65504.0)
    recon = back.astype(np.float64) + rec["lo"].view(np.float16).astype(np.float64) / 2048.0
    rel = np.abs(recon[normal] - x[normal].astype(np.float64)) / np.abs(x[normal].astype(np.float64))
    assert rel.max() <= 2.0 ** -22 * 1.0001, rel.max()
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -23.5
    # below 2^-14 the error is absolute: <= 2^-36 (before the kernel's 2^-6 input scale)
    small = ok & (np.abs(x) < 2.0 ** -14)
    assert np.abs(recon[small] - x[small].astype(np.float64)).max() <= 2.0 ** -36 * 1.0001
