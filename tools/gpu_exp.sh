#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_synthetic.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
for lib in lib_base lib; do
  for cfg in "rwkv7-2b9 Q5_1" "rwkv4-169m Q5_1"; do set -- $cfg
  RWKV_LIB_DIR=$lib timeout 300 python bench.py --config $1 --dtype $2 --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print('$1 $lib', round(d['value'],1), 'tok/s')"
  done
done; done
