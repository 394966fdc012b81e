#!/bin/bash
# round 4, step 9: comm wave takes the two odd key sets (every consumer 5 key records = all in registers), records of r/k/v/g taken during
# the x hand-over at the top of the layer (mask 255) vs without (mask 239) vs round 3
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_mega.py tests/test_gpu_real_geometry.py tests/test_gpu_abi_stream.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ) > $O/pytest_mega.txt; cat $O/pytest_mega.txt
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir, config, dtype, extra env...
  local label=$1 lib=$2 c=$3 dt=$4; shift 4
  env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype $dt --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c $dt', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
}
for rep in 1 2; do
  one base lib_base rwkv6-7b Q4_0
  one m239 lib_m239 rwkv6-7b Q4_0
  one m255 lib rwkv6-7b Q4_0
  one m255_look2 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=2
  one m255_look3 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=3
  one m255_h16 lib rwkv6-7b Q4_0 RWKV_MI_RING_HTHIN=16
done 2>&1 | tee $O/ab.txt
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -48 $O/ring_phase_trace_7b.txt; grep -n "per workgroup\|waiting for the loader" $O/ring_phase_trace_7b.txt
