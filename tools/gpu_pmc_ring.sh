#!/bin/bash
# instruction / stall counters of the persistent decode kernels (k6_ring, and k6_mega for comparison): PMC passes over a short decode
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-pmcr}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1
R=$PWD
cd /tmp
for kind in ${KINDS:-ring regs}; do
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  RWKV_MI_PERSIST=$kind timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/$O/${kind}_p$i -o p -- python $R/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 > /dev/null 2> $R/$O/${kind}_p$i.err
done
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*_p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k6_ring" in r["Kernel_Name"] or "k6_mega" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(f.split("/")[2], k, "launches", len(v), "avg", sum(v) / len(v))
PY
