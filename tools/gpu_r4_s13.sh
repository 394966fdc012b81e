#!/bin/bash
# WKV-7 sequence kernel (row_newbcast chain) A/B + stall counters of k_mmq_mfma
set -u
cd "$(dirname "$0")/.."
R=$(pwd); O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir
  env RWKV_LIB_DIR=$2 timeout 120 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 v7 prefill', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2), 'ms', flush=True)"
}
for rep in 1 2; do one prev lib_prev; one asm lib; one builtin lib_w7b; done 2>&1 | tee $O/ab_v7_prefill.txt
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7_prefill -o prefill -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_v7_prefill.err
head -8 $R/$O/prof_v7_prefill/*/*kernel_stats.csv
P="python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 --parity-tokens 0"
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$O/pmc_a -o p -- $P > /dev/null 2> $R/$O/pmc_a.err
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $R/$O/pmc_b -o p -- $P > /dev/null 2> $R/$O/pmc_b.err
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/$O/pmc_c -o p -- $P > /dev/null 2> $R/$O/pmc_c.err
timeout 120 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_INST_LEVEL_LDS --output-format csv -d $R/$O/pmc_d -o p -- $P > /dev/null 2> $R/$O/pmc_d.err
cd $R
python - <<'PY'
import csv, glob, collections
for tag in "abcd":
    fs = glob.glob(f"gpurun_out/r04n/pmc_{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k in agg:
        if "mmq_mfma" in k or "wkv6_seq" in k:
            print(tag, k, {c: round(v) for c, v in agg[k].items()})
PY
