# STFT kernel: throughput at scale + counter passes (separate runs) on the 256 x 60 s case
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3stft; mkdir -p $O; cd $R
python tools/bench_stft.py > $O/bench.txt 2>$O/bench.err && cat $O/bench.txt &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d $O/p1 -- python3 $R/tools/bench_stft.py --one > $O/p1.txt 2>$O/p1.err &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/p2 -- python3 $R/tools/bench_stft.py --one > $O/p2.txt 2>$O/p2.err &&
cd $R && python3 - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r3stft")
for d in ("p1", "p2"):
    f = glob.glob(f"{O}/{d}/*/*counter_collection.csv")[0]
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "stft_logmel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(tot): print(f"{d} {k:28s} {tot[k] / n[k]:16.0f} per launch ({n[k]} launches)")
PY
rm -rf $O/p1 $O/p2
