#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp RWKV_MI_PERSIST=ring
RWKV_MI_RING_HEAD_WG=128 timeout 120 python tools/dbg_fused.py mega-v6-4096 Q4_0 direct 2>&1 | grep RESULT
RWKV_MI_RING_HEAD_WG=192 timeout 120 python tools/dbg_fused.py mega-v6-2048 Q5_1 direct 2>&1 | grep RESULT
for w in 0 128 192 0 128 192; do
  RWKV_MI_RING_HEAD_WG=$w timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('7b HEAD_WG=$w', round(d['value'],1), round(r['avg_launch_us'],1))"; done
