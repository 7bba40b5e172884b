"""Headline benchmark: audio-seconds per wall-second of the encode()+decode() round trip (1/RTF).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): full codec inference at 24 kHz, 80 mel bins, 8 FSQ groups, BigVGAN-base
vocoder, batch 32 of 1 s clips per GPU, fp32.  A "step" is one encode() + decode(return_audios=True) over one
batch already resident in HBM; weights are seeded random (the reference ships none), audio is synthetic.
Steps are independent batches: `--lanes` (default 3) of them are in flight per GPU at a time (dmel_codec_amd/pipeline.py); the
line also carries the one-batch-at-a-time figure (`one_batch_at_a_time`) timed in the same run.
One process per GPU; utterances shard across ranks with no data-path collective (weak scaling).  Rank 0 prints
ONE JSON line with the whole-job rate, the roofline of the dominant kernel family (the implicit-GEMM convolutions:
fp32-grade products from the three-product fp16 split on the decode side and the six-product bf16 split on the
encode side, timed live with hipEvents on the launch stream) and, at N=1, the CPU baseline (the oracle restatement
made of the ATen-CPU calls the reference itself makes) timed on the host cores in the same run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA (no sparsity), same table
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(sample_rate=24000, n_mels=80, dmel_groups=8, levels=(7, 5, 5), vocoder="base_24k_100band", f_max=None),
    # the reference's own default shapes (config/codec/dMel_example.yaml + stage/pretrain.yaml): 100 mel, 10 groups
    "cfg2r": dict(sample_rate=24000, n_mels=100, dmel_groups=10, levels=(7, 5, 5), vocoder="base_24k_100band", f_max=12000.0),
}


def synth_audio(batch: int, samples: int, seed: int) -> torch.Tensor:
    """Seeded N(0,1), low-passed, peak-normalised x0.95 (mirrors dataset/lhotse_tts_dataset.py:29-32)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 1, samples, generator=g)
    k = torch.hann_window(9, periodic=False)
    x = torch.nn.functional.conv1d(x, (k / k.sum()).view(1, 1, -1), padding=4)
    return 0.95 * x / x.abs().amax(dim=-1, keepdim=True)


def build(workload: str, seed: int = 114514):
    from dmel_codec_amd.configs import build_codec
    torch.manual_seed(seed)      # seed of config/codec/dMel_example.yaml:2
    codec = build_codec(**WORKLOADS[workload])
    # BigVGAN's default init (weights ~ N(0, 0.01)) drives a random network's output to ~0; give the vocoder unit-gain
    # weights so the timed data path carries realistic magnitudes (timing does not depend on values, sin/tanh ranges do)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in codec.vocoder.named_parameters():
            if name.endswith("weight_v"):
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            elif name.endswith("weight_g"):
                p.fill_(1.0)
    return codec.eval()


def cpu_baseline(codec, workload: str, seconds_per_clip: float, budget_s: float):
    """Time the oracle (CPU restatement, kind 'port') on a bounded sample of the same workload."""
    from dmel_codec_amd.configs import oracle_cfg
    from oracle import ref_cpu
    cfg = oracle_cfg(codec)
    sd = {k: v.detach().cpu().float() for k, v in codec.state_dict().items()}
    voc = {k[len("vocoder."):]: v for k, v in sd.items() if k.startswith("vocoder.")}
    sd = {k: v for k, v in sd.items() if not k.startswith("vocoder.")}
    h = dict(codec.vocoder.h)
    n_lat = codec.decoder.input_channels
    n = 2
    L = int(cfg["sample_rate"] * seconds_per_clip)
    audio = synth_audio(n, L, 99)
    lens = torch.full((n,), L)

    def one():
        with torch.no_grad():
            ids, il = ref_cpu.vqgan_encode(sd, cfg, audio, lens)
            noise = torch.randn(n, n_lat, ids.shape[2] * 4)
            a, _ = ref_cpu.vqgan_decode(sd, cfg, ids, il, noise, voc, h)
        return a

    # The small sample does not scale to every core of a big host (oneDNN on 128 threads is slower than on 1 for
    # these shapes): calibrate one pass at a few thread counts and time the sample at the fastest.
    all_cores = torch.get_num_threads()
    one()                                    # warm-up (thread pools, oneDNN primitives)
    calib = {}
    for th in sorted({1, 8, 32, all_cores}):
        if th > all_cores:
            continue
        torch.set_num_threads(th)
        one()
        t1 = time.perf_counter()
        one()
        calib[th] = time.perf_counter() - t1
    cores = min(calib, key=calib.get)
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    reps = 0
    while True:
        one()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 20:
            break
    # one pass of the bench's own batch size (32 clips) at the same thread count: the 2-clip sample under-feeds oneDNN's batch loop
    nb = 32
    audio_b, lens_b = synth_audio(nb, L, 98), torch.full((nb,), L)
    tb = time.perf_counter()
    with torch.no_grad():
        ids_b, il_b = ref_cpu.vqgan_encode(sd, cfg, audio_b, lens_b)
        ref_cpu.vqgan_decode(sd, cfg, ids_b, il_b, torch.randn(nb, n_lat, ids_b.shape[2] * 4), voc, h)
    batch32 = nb * seconds_per_clip / (time.perf_counter() - tb)
    torch.set_num_threads(all_cores)
    return {"value": round(n * seconds_per_clip * reps / el, 3), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "one_thread_audio_sec_per_sec": round(n * seconds_per_clip / calib[1], 3),
            "batch32_one_pass_audio_sec_per_sec": round(batch32, 3),
            "sample": f"{reps} x (encode+decode of {n} x {seconds_per_clip:g} s clips) through oracle/ref_cpu.py "
                      f"(torch.stft / F.conv1d / F.conv_transpose1d), {el:.1f} s of CPU work on {cores} threads "
                      f"(fastest of a one-pass calibration over {sorted(calib)} threads; host has {all_cores}); plus one pass of "
                      f"{nb} clips on the same threads and the 1-thread figure of the calibration",
            "one_pass_audio_sec_per_sec_by_threads": {str(k): round(n * seconds_per_clip / v, 3) for k, v in calib.items()}}


def committed_traffic(conv=None):
    """HBM bytes per launch of the dominant kernel from the PMC passes (rocprofv3 --pmc, separate runs, corrected as the MI355X guide
    prescribes).  bench.py cannot read hardware counters itself: the figure is the one committed with the profile it came from
    (profiles/pmc_traffic.json: {"bytes_per_launch": ..., "source": ...}); null when no such file travels with the repo."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        alg = d.get("algorithmic_bytes_per_launch")
        if conv is not None and conv.get("launches"):
            alg = round(conv["bytes"] / conv["launches"])        # counted live by the library for exactly these launches
        now = kernel_source_hash()
        return {"bytes_per_launch": d["bytes_per_launch"], "algorithmic_bytes_per_launch": alg,
                "ratio": round(d["bytes_per_launch"] / alg, 3) if alg else None,
                "measured_in": "committed profile (profiles/pmc_traffic.json), NOT this run: bench.py cannot read hardware counters",
                "kernel_source_hash_of_profile": d.get("kernel_source_hash"), "kernel_source_hash_now": now,
                "stale": d.get("kernel_source_hash") != now,      # true: the convolution kernels changed since the PMC passes were taken
                "source": d.get("source")}
    except (OSError, KeyError, ValueError):
        return None


def kernel_source_hash() -> str:
    """sha256 (first 16 hex digits) over the convolution kernel sources: ties a PMC figure to the code it was measured on."""
    import hashlib
    hsh = hashlib.sha256()
    for name in ("conv_igemm.hip", "conv_dev.h", "conv.h", "wavenet_fused.hip"):
        try:
            with open(os.path.join(ROOT, "dmel_codec_amd", "csrc", name), "rb") as f:
                hsh.update(f.read())
        except OSError:
            pass
    return hsh.hexdigest()[:16]


def visible_gpu_count() -> int:
    """Number of GPUs a child rank will see, WITHOUT any torch.cuda / HIP call in this process (the launcher parent must provably never
    initialise the GPU before it starts children): KFD topology nodes with a non-zero gfx_target_version, cut down by the
    *_VISIBLE_DEVICES variables the runtime honours."""
    import glob
    n = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(prop) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k == "gfx_target_version" and int(v.strip() or 0) != 0:
                        n += 1
        except (OSError, ValueError):
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def per_rank_times(dist, elapsed: float, device):
    """Every rank's own wall time of the timed region (a straggler shows up here, not only in the max)."""
    if dist is None:
        return [elapsed]
    t = torch.zeros(dist.get_world_size(), device=device, dtype=torch.float64)
    t[dist.get_rank()] = elapsed
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def max_over_ranks(dist, elapsed: float, device) -> float:
    """Timing contract: the step time of the job is the slowest rank's."""
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_rate(world: int, batch: int, seconds: float, steps: int, elapsed: float) -> float:
    """Whole-job audio-seconds per wall-second: every rank processed its own `batch` clips per step (weak scaling)."""
    return world * batch * seconds * steps / elapsed


def spawn_ranks(n: int, argv, script: str | None = None, have: int | None = None) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: become the launcher.  The N ranks are CHILD processes started before
    this process has touched the GPU (the devices are counted from the KFD topology in sysfs, not through torch.cuda; nothing here execs),
    one per device, wired with the same RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables torch.distributed.run sets.  Rank 0's
    stdout (the one JSON line) is forwarded; the exit code is non-zero if any rank fails."""
    import socket
    import subprocess
    have = visible_gpu_count() if have is None else have
    if have < n:
        print(f"bench.py --gpus {n} needs {n} devices on this node, found {have}", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                code = pr.poll()
                if code is None:
                    continue
                pending.remove(pr)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for other in pending:                  # a failed rank leaves the others in a collective: stop them
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step")
    ap.add_argument("--seconds", type=float, default=1.0, help="clip length")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work for the baseline (0 = skip)")
    ap.add_argument("--median-steps", type=int, default=100,
                    help="extra steps timed one by one with events for the median / p10 / p90 of the step time (0 = skip)")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 3),
                    help="streams the three AMP blocks of a BigVGAN stage run on WITHIN one batch (3 = overlapped: the round-1/2 default; with "
                         "several batches in flight the lanes already fill the chip and 1 measures faster: profiles/r03_lanes_sweep.txt)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight per GPU in the timed region (dmel_codec_amd.pipeline.CodecLanes: replicas of the codec on their "
                         "own streams, batches dealt round-robin; 1 = one batch at a time, the round-1/2 form, also timed and reported)")
    ap.add_argument("--decode-precision", default="fp32", choices=("fp32", "fp32_bf16x3", "bf16"),
                    help="fp32 (default, the parity path and the headline number: fp32-grade products, three-product fp16 split on the "
                         "decode side), fp32_bf16x3 (six-product bf16 split everywhere: the round-1 arithmetic, for A/B) or the opt-in "
                         "bf16-operand mode of the decode-side convolutions (never the default: it is outside the 1e-4 bar)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))     # plain `python bench.py --gpus N`: start the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DMEL_BENCH_FORCE_DIST") == "1":      # the second: tests drive the RCCL control path with one rank
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev)    # RCCL
        dist = dist_mod

    from dmel_codec_amd import _lib
    codec = build(args.workload)
    cpu_base = None
    if world == 1 and rank == 0 and args.cpu_budget > 0:
        # the host-core baseline runs first, on the freshly built (CPU-resident) weights, so the GPU legs close the run
        cpu_base = cpu_baseline(codec, args.workload, args.seconds, args.cpu_budget)
    codec = codec.to(dev)
    sr = WORKLOADS[args.workload]["sample_rate"]
    L = int(sr * args.seconds)
    audio = synth_audio(args.batch, L, 1234 + rank).to(dev)      # every rank its own utterances
    lens = torch.full((args.batch,), L, device=dev, dtype=torch.int64)

    def step():
        ids, il = codec.encode(audio, lens)
        wav, _ = codec.decode(ids, il, return_audios=True)       # draws its Gaussian noise like the reference (:473)
        return ids, wav

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def configure(c):
        c.set_decode_precision(args.decode_precision)
        c.vocoder.set_streams(args.streams)

    # One step = one batch through encode() + decode().  Batches are independent, so `--lanes` of them are in flight at a time: every lane
    # is a replica of the codec with its own workspaces and stream (dmel_codec_amd/pipeline.py), steps are dealt to the lanes round-robin,
    # and the WaveNet phase of one batch (few, long workgroups) runs under the vocoder phase of another.  Every step's work is complete
    # when the timed region ends (device-wide synchronise); nothing is shared or skipped between steps.
    from dmel_codec_amd.pipeline import CodecLanes
    lanes = CodecLanes(codec, max(1, args.lanes))
    lanes.configure(configure)

    def timed(fn):
        for _ in range(max(args.warmup, len(lanes))):
            fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        sync_all()
        return time.perf_counter() - t0

    last = {}

    def one_step():
        last["ids"], last["wav"] = step()

    def lane_step():
        last["r"] = lanes.roundtrip(audio, lens)

    elapsed_one_lane = timed(one_step) if len(lanes) > 1 else None      # one batch at a time (what rounds 1 and 2 reported), for the record
    elapsed = timed(lane_step if len(lanes) > 1 else one_step)
    rank_elapsed = per_rank_times(dist, elapsed, dev)
    elapsed = max_over_ranks(dist, elapsed, dev)
    lanes_ids_ok = None
    if elapsed_one_lane is not None:
        elapsed_one_lane = max_over_ranks(dist, elapsed_one_lane, dev)
        ids, _, wav = last["r"].wait()
        # the token ids do not depend on the decoder's noise: the batch that went through a lane with others in flight must give exactly the
        # ids of the same batch on its own (a kernel-level interference between lanes would show here: profiles/r03_stft_concurrency.txt)
        torch.cuda.synchronize()
        lanes_ids_ok = bool(torch.equal(ids, last["ids"]))
        if not lanes_ids_ok:
            raise SystemExit("bench.py: token ids of a batch processed with other batches in flight differ from the same batch alone")
    else:
        ids, wav = last["ids"], last["wav"]

    # Distribution of the step time: `median_steps` further steps ONE BATCH AT A TIME, each bracketed by its own pair of events on the launch
    # stream (the vocoder's side streams fork from and join back into it, so the pair sees the whole step): the latency of a batch.
    per_step_ms = []
    if args.median_steps > 0:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.median_steps)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        sync_all()
        per_step_ms = sorted(a.elapsed_time(b) for a, b in evs)

    # Per-kernel roofline pass.  In the timed region above BigVGAN's three AMP blocks run on three streams, so kernels
    # overlap and a kernel's own hipEvent interval no longer measures that kernel alone.  The same K steps are therefore
    # repeated with the vocoder serialised onto one stream and every launch bracketed by hipEvents on its stream.
    codec.vocoder.set_streams(1)
    step()
    sync_all()
    _lib.prof_reset()
    _lib.prof_enable(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed_serial = time.perf_counter() - t1
    _lib.prof_enable(False)
    codec.vocoder.set_streams(args.streams)
    assert torch.isfinite(wav).all() and wav.shape == (args.batch, 1, (L // 256 // 4) * 4 * 256)

    conv = _lib.prof_read("conv_igemm")
    snake = _lib.prof_read("aa_snake")
    stft = _lib.prof_read("stft_logmel")
    small = _lib.prof_read("small")
    _lib.prof_reset()

    if rank == 0:
        rate = job_rate(world, args.batch, args.seconds, args.steps, elapsed)
        ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
        bf16 = args.decode_precision == "bf16"
        native = os.environ.get("DMEL_CONV_FP32_MFMA", "0") not in ("", "0")
        # Matrix-core ceiling of the family: every launch reports the flops it ISSUES for its algorithmic fp32 flops (6x under the three-way
        # bf16 split: encoder, quantiser; 3x under the two-way fp16 split: decoder WaveNet, vocoder), so the ceiling
        # for the step's ALGORITHMIC flops is the dense 16-bit MFMA peak divided by the flop-weighted mean of those factors.
        factor = conv["issue_flops"] / conv["flops"] if conv["flops"] > 0 else 6.0
        peak = PEAK_BF16_MFMA_TFLOPS / factor
        if bf16:
            kernel = "conv_bf16_kernel<NP=1> (v_mfma_f32_32x32x16_bf16) decode side + six-product split on the encoder side"
        elif native:
            kernel = "conv_igemm_kernel (fp32 v_mfma_f32_32x32x2_f32)"
        elif args.decode_precision == "fp32_bf16x3":
            kernel = ("conv_bf16_kernel<NP=3> (fp32 via exact 3-way bf16 operand split: 6 x v_mfma_f32_32x32x16_bf16 per 32x32x16 "
                      "block, fp32 accumulate)")
        else:
            kernel = ("conv_bf16_kernel<NP=2> on the decode side (fp32 via 2-way fp16 operand split, 22 significant bits per operand as in 3xTF32: 3 x "
                      "v_mfma_f32_32x32x16_f16 per 32x32x16 block, two fp32 accumulators; inputs staged x 2^-6 with a FIXED scale: domain "
                      "|x| < 4.19e6, absolute error floor 2^-30 below |x| = 2^-8, include/dmel_hip.h) + <NP=3> / wavenet_fused_kernel on the "
                      "encode side (3-way bf16 split, 6 MFMAs per block: the ids are defined by it)")
        out = {
            "metric": "audio-sec/sec (encode+decode RTF) @24 kHz batch 32",
            "value": round(rate, 2),
            "unit": "audio-sec/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_step_by_rank": [round(1e3 * e / args.steps, 3) for e in rank_elapsed],
            "ms_per_step_events": ({"what": "latency of one batch alone (no other batch in flight)", "n": len(per_step_ms), "median": round(per_step_ms[len(per_step_ms) // 2], 3),
                                    "p10": round(per_step_ms[len(per_step_ms) // 10], 3),
                                    "p90": round(per_step_ms[(9 * len(per_step_ms)) // 10], 3)} if per_step_ms else None),
            "process_group": (dist.get_backend() if dist is not None else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands, f32 accumulate in the decode convolutions (opt-in mode); f32 elsewhere" if bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: encode+decode, {sr} Hz, {WORKLOADS[args.workload]['n_mels']} mel, "
                                   f"{WORKLOADS[args.workload]['dmel_groups']} FSQ groups {list(WORKLOADS[args.workload]['levels'])}, "
                                   f"WaveNet 20+20 layers, BigVGAN-base, batch {args.batch} x {args.seconds:g} s per GPU",
                       "parallelism": f"{world} x independent utterance shards, no collective; {len(lanes)} independent batch(es) in flight "
                                      "per GPU (dmel_codec_amd.pipeline.CodecLanes)",
                       "lanes": len(lanes), "vocoder_streams": args.streams, "lane_ids_equal_single_batch_ids": lanes_ids_ok},
            "one_batch_at_a_time": ({"value": round(job_rate(world, args.batch, args.seconds, args.steps, elapsed_one_lane), 2),
                                     "ms_per_step": round(1e3 * elapsed_one_lane / args.steps, 3)} if elapsed_one_lane is not None else None),
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": committed_traffic(conv),
                         "mfma_flops_issued_per_algorithmic_flop": round(factor, 3),
                         # the same rate against the ceiling of rounds 1 / early 2, when every convolution issued six products (2500 / 6)
                         "frac_of_six_product_ceiling": round(ach / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4),
                         "frac_of_fp32_mfma_peak": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                         "launches_per_step": conv["launches"] // max(1, args.steps),
                         "avg_launch_us": round(1e3 * conv["ms"] / max(1, conv["launches"]), 2),
                         "gflop_per_step": round(conv["flops"] / args.steps / 1e9, 1),
                         "measured_in": "serialised re-run of the same steps (one batch at a time, single stream, hipEvents around every launch); "
                                        "the timed region overlaps kernels of several batches and streams",
                         "ms_per_step_serialised": round(1e3 * elapsed_serial / args.steps, 3)},
            "kernel_ms_per_step": {"conv_igemm": round(conv["ms"] / args.steps, 3), "aa_snake": round(snake["ms"] / args.steps, 3),
                                   "stft_logmel": round(stft["ms"] / args.steps, 4), "small": round(small["ms"] / args.steps, 3)},
            "aa_snake_hbm": {"achieved_GBs": round(snake["bytes"] / (snake["ms"] * 1e-3) / 1e9, 1) if snake["ms"] > 0 else 0.0,
                             "peak_GBs": PEAK_HBM_GBS},
        }
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
