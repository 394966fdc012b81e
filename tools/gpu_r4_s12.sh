#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_mega.py tests/test_gpu_real_geometry.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 ) > $O/pytest_mega.txt; cat $O/pytest_mega.txt
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir, config, dtype, extra env...
  local label=$1 lib=$2 c=$3 dt=$4; shift 4
  env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype $dt --steps 256 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c $dt', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
}
for rep in 1 2 3; do one base lib_base rwkv6-1b6 Q4_0; one final lib rwkv6-1b6 Q4_0; done 2>&1 | tee $O/ab.txt
for rep in 1 2; do one base lib_base rwkv6-7b Q4_0; one final lib rwkv6-7b Q4_0; done 2>&1 | tee -a $O/ab.txt
rm -f /tmp/synthetic-rwkv6-7b-Q4_0*
for rep in 1 2; do one base lib_base rwkv6-7b Q8_0; one final lib rwkv6-7b Q8_0; done 2>&1 | tee -a $O/ab.txt
rm -f /tmp/synthetic-rwkv6-7b*
for rep in 1 2; do one base lib_base rwkv6-7b Q5_1; one final lib rwkv6-7b Q5_1; done 2>&1 | tee -a $O/ab.txt
