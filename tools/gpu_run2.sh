#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_mul_mat.py -m gpu -q -x --timeout 600 2>&1 | tail -30 ) > $O/pytest_prefill.txt
tail -5 $O/pytest_prefill.txt
timeout 600 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 6 > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err
tail -c 1500 $O/prefill_1b6_q4_0.json; tail -3 $O/prefill_1b6.err
