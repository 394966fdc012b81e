#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_pipeline_cpp.py -x -q -m gpu 2>&1 | tail -5
for v in 0 1; do
  if [ $v = 1 ]; then export RWKV_MI_NO_WKV7_SEQ=1; else unset RWKV_MI_NO_WKV7_SEQ; fi
  timeout 600 python bench.py --mode prefill --config rwkv7-2b9 --dtype Q5_1 --steps 3 --warmup 1 --parity-tokens 64 --cpu-seconds 5 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('NO_SEQ=$v', d['value'], 'tok/s', d['ms_per_step'], 'ms', d.get('parity'))"
done
