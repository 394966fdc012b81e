#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r02c}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_prefill.py -m gpu -q -x -k "sequence_pass or real_head" 2>&1 | tail -4 ) > $O/pytest.txt; cat $O/pytest.txt
timeout 300 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 4 > $O/prefill.json 2> $O/prefill.err
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o prefill -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/rocprof.err
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 "$f" | cut -c1-150
python - $O/prefill.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d.get("parity"))
PY
