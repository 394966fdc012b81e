#!/bin/bash
# decode A/B of two library builds: tools/gpu_r4_s17.sh <config> <dtype> <libA> <libB>
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
C=$1; DT=$2; shift 2
one() {
  env RWKV_LIB_DIR=$1 timeout 100 python bench.py --config $C --dtype $DT --steps 256 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1 $C $DT', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),2), 'us', flush=True)"
}
for rep in 1 2; do for L in "$@"; do one $L; done; done 2>&1 | tee $O/ab_$C.txt
