#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export RWKV_MI_PERSIST=ring
for a in "mega-v6-2048 Q4_0" "mega-v6-4096 Q4_0" "mega-v6-4096 Q5_1" "mega-v6-2048-v8k Q8_0" "mega-v6-2048 Q4_1" "mega-v6-4096 Q8_0"; do timeout 120 python tools/dbg_fused.py $a direct 2>&1 | grep RESULT; done
for cfg in rwkv6-7b rwkv6-1b6; do
for d in 0 64 0 64; do
  RWKV_MI_RING_DBG=$d timeout 300 python bench.py --config $cfg --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg DBG=$d', round(d['value'],1), 'tok/s kernel', round(d['roofline'].get('avg_launch_us'),1), round(d['roofline']['frac'],4))"
done; done
RWKV_MI_PERSIST=ring timeout 200 python tools/trace_ring.py rwkv6-7b 5 2>&1 | grep -E "per workgroup|by quarter|^(A\.|C\.|E\.|F\.|G\.)|layer wall|waiting for the loader" | head -40
