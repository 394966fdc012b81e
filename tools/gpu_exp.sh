#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export RWKV_MI_PERSIST=ring
RWKV_MI_RING_PFW=64 timeout 120 python tools/dbg_fused.py mega-v6-4096 Q4_0 direct 2>&1 | grep RESULT
RWKV_MI_RING_PFW=64 timeout 120 python tools/dbg_fused.py mega-v6-2048-v8k Q5_1 direct 2>&1 | grep RESULT
for w in 0 32 64 128 0 64; do
  RWKV_MI_RING_PFW=$w timeout 300 python bench.py --config rwkv6-7b --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('7b PFW=$w', round(d['value'],1), 'tok/s kernel', round(d['roofline'].get('avg_launch_us'),1), round(d['roofline']['frac'],4))"
done
RWKV_MI_RING_PFW=64 timeout 200 python tools/trace_ring.py rwkv6-7b 5 2>&1 | grep -E "per workgroup|^(A\.|C\.|E\.|F\.|G\.)|layer wall|waiting for the loader|LOADER:" | head -30
