#!/bin/bash
# k_mmq_mfma: where the time goes -- timing-only variants (results invalid): no chunk sync, no DMA, no fold
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04q; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
one() {
  env RWKV_LIB_DIR=$2 timeout 100 python bench.py --config ${3:-rwkv6-1b6} --dtype ${4:-Q4_0} --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', round(d['value'],1), 'tok/s', round(d['ms_per_step'],2), 'ms; gemm', round(r['avg_launch_us'],2), 'us x', r['launches'], flush=True)"
}
for L in "$@"; do one $L $L; done 2>&1 | tee $O/variants.txt
