#!/bin/bash
# the loader alone inside the real kernel (RWKV_MI_RING_DBG=8): streaming rate per CU for a few in-flight depths
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_MI_NO_GRAPH=1
for k in "$@"; do
  env $k RWKV_MI_RING_DBG=8 timeout 300 python bench.py --config rwkv6-7b --dtype Q4_0 --steps 32 --warmup 4 --cpu-seconds 0 --abi-tokens 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); us=r.get('avg_launch_us',0); print('$k loader alone: kernel', round(us,1), 'us ->', round(4.21e9/256/us/1e3,1) if us else 0, 'GB/s per CU,', round(4.21e9/us/1e6,2) if us else 0, 'TB/s chip')"
done
