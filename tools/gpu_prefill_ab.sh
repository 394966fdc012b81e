#!/bin/bash
# parity tests (GEMM, sequence pass, serial == sequence), then same-box prefill bench: lib_prev vs lib
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
mkdir -p gpurun_out/pfab
timeout 900 python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_prefill.py tests/test_gpu_reference_programs.py tests/test_gpu_tiny_rwkv.py tests/test_gpu_synthetic.py tests/test_gpu_api_semantics.py -x -q -m gpu 2>&1 | tail -15
for v in lib_prev lib lib_prev lib; do
  RWKV_LIB_DIR=$v timeout 300 python bench.py --mode prefill --config rwkv6-1b6 --dtype Q4_0 --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/pfab/$v.json
  python -c "
import json; d=json.load(open('gpurun_out/pfab/$v.json')); r=d['roofline']
print('$v prefill', round(d['value'],1), 'tok/s;', round(d['ms_per_step'],2), 'ms; gemm', round(r['achieved'],1), 'TOP/s avg', round(r['avg_launch_us'],1), 'us; parity', (d.get('parity') or {}).get('equal'))"
done
for v in lib; do echo "== $v"; RWKV_LIB_DIR=$v timeout 200 python tools/gemm_bench.py Q4_0 2>&1 | grep time_mm | sed -n "4p;7p"; done
