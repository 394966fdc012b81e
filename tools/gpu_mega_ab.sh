#!/bin/bash
# same-box A/B of the persistent kernel: lib_base (reference build) vs lib (current), alternating, 7B and 1.6B
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1
for a in "mega-v6-2048 Q4_0 direct" "mega-v6-4096 Q4_0 direct"; do timeout 100 python tools/dbg_fused.py $a 2>&1 | grep -E "RESULT" ; done
for rep in 1 2; do for v in lib_base lib; do for c in rwkv6-7b rwkv6-1b6; do
RWKV_LIB_DIR=$v timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$v $c', round(d['value'],1), 'tok/s; mega', round(r.get('avg_launch_us',0),1), 'us')"
done; done; done
