#!/bin/bash
# k_wkv7_seq (out chain one step behind, ring entry read a step ahead, whole-chunk loop): v7 sequence tests + prefill line
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
( timeout 200 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_real_geometry.py -m gpu -q -x -p no:cacheprovider -k "wkv7 or v7 or rwkv7" 2>&1 | tail -3 ) > $O/pytest_v7.txt; cat $O/pytest_v7.txt
one() {
  env RWKV_LIB_DIR=$1 timeout 100 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 v7 prefill', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2), 'ms', flush=True)"
}
for rep in 1 2; do one lib_prev; one lib; done 2>&1 | tee $O/ab_v7_prefill.txt
