#!/bin/bash
# stall attribution of the sequence-mode GEMM: PMC passes over tools/gemm_one.py (one shape, a few launches)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-pmcg}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/$O/p$i -o p -- python $R/tools/gemm_one.py ${GEMM_SHAPE:-2048 2048 1024} > /dev/null 2> $R/$O/p$i.err
  tail -2 $R/$O/p$i.err
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mmq_mfma" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(f.split("/")[2], k, "launches", len(v), "avg", sum(v) / len(v))
PY
