#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_fused.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for lib in lib_base lib; do
  RWKV_LIB_DIR=$lib timeout 300 python bench.py --config rwkv7-2b9 --dtype Q5_1 --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('v7 $lib', round(d['value'],1), 'tok/s', r.get('kernel'), round(r.get('avg_launch_us',0),1))"
done; done
