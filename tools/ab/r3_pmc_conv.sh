#!/bin/bash
# counter passes over three convolution shapes (each pass its own run, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3/pmc_conv; mkdir -p $OUT
run() { # name, counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -- python3 tools/bench_conv.py --precision 3 --iters 3 --only "$SHAPES" > $OUT/$n.log 2>&1
}
for SHAPES in wn_dec_k3 bv2_k7 bv4_k3; do
  run ${SHAPES}_A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
  run ${SHAPES}_B SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  run ${SHAPES}_C TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run ${SHAPES}_D SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
done
python3 - <<'PY'
import csv, glob, collections, os
base="gpurun_out/r3/pmc_conv"
for d in sorted(glob.glob(base+"/*_[ABCD]")):
    f=glob.glob(d+"/*/*counter_collection.csv")
    if not f: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "conv_bf16_kernel" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(os.path.basename(d), k)
        for c,x in sorted(v.items()):
            print(f"      {c:28s} {sum(x)/len(x):16.0f}")
PY
