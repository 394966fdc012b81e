#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp RWKV_MI_PERSIST=ring RWKV_MI_NO_AUTOTUNE=1
for a in "mega-v6-2048 Q4_0" "mega-v6-4096 Q4_0" "mega-v6-4096 Q5_1" "mega-v6-2048-v8k Q8_0"; do timeout 120 python tools/dbg_fused.py $a direct 2>&1 | grep RESULT; done
for rep in 1 2 3; do for lib in lib_base lib; do
  RWKV_LIB_DIR=$lib timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('7b $lib', round(d['value'],1), round(r['avg_launch_us'],1))"; done; done
