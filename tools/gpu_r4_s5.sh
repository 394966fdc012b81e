#!/bin/bash
# round 4, step 5: + ffn receptance records taken behind the key records during the yq hand-over (mask 239) vs mask 111 vs round 3
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mega.py tests/test_gpu_real_geometry.py -m gpu -q -x -p no:cacheprovider -k "matches_oracle or wrap or world" 2>&1 | tail -5 ) > $O/pytest_mega.txt; cat $O/pytest_mega.txt
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1
one() {  # label, lib dir, config, dtype, extra env...
  local label=$1 lib=$2 c=$3 dt=$4; shift 4
  env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype $dt --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c $dt', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
}
for rep in 1 2; do
  one base lib_base rwkv6-7b Q4_0
  one m111 lib_m111 rwkv6-7b Q4_0
  one m239 lib rwkv6-7b Q4_0
  one m239_nap0 lib rwkv6-7b Q4_0 RWKV_MI_RING_NAP=0
  one m239_thin32 lib rwkv6-7b Q4_0 RWKV_MI_RING_THIN=32
  one m239_thin48 lib rwkv6-7b Q4_0 RWKV_MI_RING_THIN=48
done 2>&1 | tee $O/ab.txt
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -48 $O/ring_phase_trace_7b.txt
rm -f /tmp/synthetic-rwkv6-7b-Q4_0*
for rep in 1 2; do one base lib_base rwkv6-7b Q8_0; one m239 lib rwkv6-7b Q8_0; done 2>&1 | tee -a $O/ab.txt
