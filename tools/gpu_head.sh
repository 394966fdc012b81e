#!/bin/bash
# head phase of the ring kernel: stamps + A/B against the separate head launch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; T=${1:-h}
export RWKV_MI_PERSIST=ring
timeout 300 python tools/trace_head.py rwkv6-7b > gpurun_out/head_trace_$T.txt 2>&1
for v in 0 1; do
  RWKV_MI_RING_NO_HEAD=$v timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('NO_HEAD=$v', d['value'], 'tok/s kernel', d['roofline'].get('avg_launch_us'), d['roofline']['frac'])" >> gpurun_out/head_trace_$T.txt 2>&1
done
cat gpurun_out/head_trace_$T.txt
