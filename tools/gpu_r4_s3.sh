#!/bin/bash
# round 4, step 3: loader depth while waves watch hand-overs (RWKV_MI_RING_HTHIN) x normal depth, same-box against the round-3 kernel
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mega.py -m gpu -q -x -p no:cacheprovider -k "matches_oracle or wrap" 2>&1 | tail -5 ) > $O/pytest_mega.txt; cat $O/pytest_mega.txt
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir, config, extra env...
  local label=$1 lib=$2 c=$3; shift 3
  env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
}
for rep in 1 2; do
  one base lib_base rwkv6-7b
  one new lib rwkv6-7b
  one new_h16 lib rwkv6-7b RWKV_MI_RING_HTHIN=16
  one new_h8 lib rwkv6-7b RWKV_MI_RING_HTHIN=8
  one new_h24 lib rwkv6-7b RWKV_MI_RING_HTHIN=24
  one new_i24_h16 lib rwkv6-7b RWKV_MI_RING_INFLIGHT=24 RWKV_MI_RING_HTHIN=16
  one new_i16 lib rwkv6-7b RWKV_MI_RING_INFLIGHT=16 RWKV_MI_RING_HTHIN=16
  one new_i32_h16_t8 lib rwkv6-7b RWKV_MI_RING_INFLIGHT=32 RWKV_MI_RING_HTHIN=16 RWKV_MI_RING_THIN=8
  one new_h16_nap0 lib rwkv6-7b RWKV_MI_RING_HTHIN=16 RWKV_MI_RING_NAP=0
  one new_h16_nap6 lib rwkv6-7b RWKV_MI_RING_HTHIN=16 RWKV_MI_RING_NAP=6
  one new_nopre lib rwkv6-7b RWKV_MI_RING_DBG=64
done 2>&1 | tee $O/ab.txt
one base lib_base rwkv6-1b6; one new lib rwkv6-1b6; one base lib_base rwkv6-1b6; one new lib rwkv6-1b6
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -48 $O/ring_phase_trace_7b.txt
RWKV_MI_RING_HTHIN=16 timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b_h16.txt 2>> $O/trace.err; head -48 $O/ring_phase_trace_7b_h16.txt
