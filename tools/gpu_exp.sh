#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp RWKV_MI_PERSIST=ring RWKV_MI_NO_AUTOTUNE=1
run() { env "$@" timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', round(d['value'],1), round(r['avg_launch_us'],1))"; }
run A=0; run RWKV_MI_RING_INFLIGHT=40; run RWKV_MI_RING_INFLIGHT=52; run RWKV_MI_RING_THIN=8; run RWKV_MI_RING_THIN=24; run A=0
run RWKV_MI_RING_NAP=1; run RWKV_MI_RING_NAP=4; run RWKV_MI_RING_BURST=16; run RWKV_MI_RING_BURST=8; run A=0
