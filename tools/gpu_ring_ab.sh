#!/bin/bash
# same-box A/B of two builds of the ring kernel: lib_base (make LIBDIR=lib_base OBJDIR=build_base of the reference commit) vs lib, alternating
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
for a in "mega-v6-2048 Q4_0 direct" "mega-v6-4096 Q4_0 direct" "mega-v6-4096 Q8_0 direct" "mega-v6-2048 Q5_1 direct"; do timeout 100 python tools/dbg_fused.py $a 2>&1 | grep -E "RESULT"; done
for rep in 1 2 3; do for v in lib_base lib; do for c in rwkv6-7b ${AB_CONFIGS:-}; do
RWKV_LIB_DIR=$v timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$v $c', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us')"
done; done; done
RWKV_MI_PERSIST=regs timeout 300 python bench.py --config rwkv6-7b --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('regs rwkv6-7b', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us')"
