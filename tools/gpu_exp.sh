#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp
timeout 300 python -m pytest tests/test_gpu_mega.py -q -x -m gpu -p no:cacheprovider -k "device_side or new_context or concurrent" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('7b', round(d['value'],1), 'kind', d['config']['persist_kind'], round(r['avg_launch_us'],1), round(r['frac'],4), 'traffic', r.get('traffic'), str(r.get('traffic_source'))[:50])"; done
