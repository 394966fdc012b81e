#!/bin/bash
# Third evidence run of round 4 (k_wkv7_seq's out chain a step behind): the GPU suite + the RWKV-7 prefill line and its kernel stats.
set -u
cd "$(dirname "$0")/.."
T=${1:-r04x}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
( timeout 330 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error" | tail -8 ) > $O/pytest.txt; cat $O/pytest.txt
timeout 120 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 6 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7_prefill -o prefill -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_v7_prefill.err
cd $R
python - "$O/prefill_7v_2b9_q5_1.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "parity", (d.get("parity") or {}).get("equal"))
PY
head -4 $O/prof_v7_prefill/prefill_kernel_stats.csv | cut -c1-60,150-220
