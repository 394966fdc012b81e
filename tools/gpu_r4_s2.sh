#!/bin/bash
# round 4, step 2: the ring kernel with every record of a phase taken into registers during the hand-over wait -- parity on both
# geometries / all formats, then a same-box A/B against the round-3 kernel (rwkv.cpp_amd/lib_base), and the phase trace
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_mega.py tests/test_gpu_real_geometry.py -m gpu -q -x -p no:cacheprovider -k "mega or world" 2>&1 | tail -15 ) > $O/pytest_mega.txt; cat $O/pytest_mega.txt
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir, extra env...
  local label=$1 lib=$2; shift 2
  for c in rwkv6-7b; do
    env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
  done
}
for rep in 1 2; do
  one base lib_base
  one new lib
  one new_nopre lib RWKV_MI_RING_DBG=64
  one new_thin8 lib RWKV_MI_RING_THIN=8
  one new_thin32 lib RWKV_MI_RING_THIN=32
  one new_infl32 lib RWKV_MI_RING_INFLIGHT=32
done 2>&1 | tee $O/ab.txt
RWKV_LIB_DIR=lib timeout 200 python bench.py --config rwkv6-1b6 --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | tail -c 600
RWKV_LIB_DIR=lib_base timeout 200 python bench.py --config rwkv6-1b6 --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | tail -c 600
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -45 $O/ring_phase_trace_7b.txt
