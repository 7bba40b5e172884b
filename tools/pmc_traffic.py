"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes) of
`bench.py --streams 1` into profiles/pmc_traffic.json: HBM-side bytes per launch of the dominant kernel family (the split-fp32
implicit-GEMM convolution, including the whole-stack WaveNet kernel), with the guide's gfx950 correction (FETCH_SIZE counts half of
the bytes of a wide coalesced read: doubled; WRITE_SIZE exact; counter unit KB).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import defaultdict


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] += float(row["Counter_Value"])
            n[name] += 1
    return tot, n


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    ft, fn = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wn = per_kernel(write_csv, "WRITE_SIZE")
    fam = lambda k: "conv_bf16_kernel" in k or "conv_pc_kernel" in k or "conv_snake_kernel" in k or "conv_igemm_kernel" in k or "conv_direct_kernel" in k or "wavenet_fused_kernel" in k
    launches = sum(v for k, v in fn.items() if fam(k))
    fetch_kb = sum(v for k, v in ft.items() if fam(k))
    write_kb = sum(v for k, v in wt.items() if fam(k))
    launches_w = sum(v for k, v in wn.items() if fam(k))
    table = {}
    for k in sorted(ft, key=lambda k: -ft[k])[:12]:
        table[k[:90]] = {"launches": fn[k], "fetch_raw_MB_per_launch": round(ft[k] / fn[k] / 1024, 2),
                         "write_MB_per_launch": round(wt.get(k, 0.0) / max(1, wn.get(k, 1)) / 1024, 2)}
    res = {"bytes_per_launch": round((2.0 * fetch_kb / launches + write_kb / max(1, launches_w)) * 1024),
           "reads_corrected_bytes_per_launch": round(2.0 * fetch_kb / launches * 1024),
           "writes_bytes_per_launch": round(write_kb / max(1, launches_w) * 1024),
           "launches_in_profile": launches,
           "algorithmic_bytes_per_launch": None,
           "kernel_source_hash": __import__("bench").kernel_source_hash(),     # bench.py flags the figure as stale when the kernels change
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) over `bench.py --steps 2 --warmup 1 --cpu-budget 0 "
                     "--median-steps 0 --streams 1`; conv family = conv_bf16_kernel<...> + conv_pc_kernel<...> + wavenet_fused_kernel; FETCH_SIZE doubled "
                     "(gfx950 reports half of a wide coalesced read), WRITE_SIZE exact, KB -> bytes; profiles/r03_pmc_*.csv",
           "top_kernels": table}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "top_kernels"}))


if __name__ == "__main__":
    main()
