#!/bin/bash
# quick check of the persistent kernel: bit-exactness on the two geometries (dbg_fused.py) + bench of the 7B and the 1.6B
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1
for a in "mega-v6-2048 Q4_0 direct" "mega-v6-4096 Q4_0 direct" "mega-v6-4096 Q5_1 direct" "mega-v6-2048 Q8_0 direct"; do timeout 100 python tools/dbg_fused.py $a 2>&1 | grep -E "RESULT|path" | tr '\n' ' '; echo; done
for c in rwkv6-7b rwkv6-1b6; do
timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$c', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms; mega', round(r.get('avg_launch_us',0),1), 'us frac', round(r.get('frac',0),4))"
done
