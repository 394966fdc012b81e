#!/bin/bash
# round 4, step 6: takes per sentinel look (RWKV_MI_RING_LOOK) + the F16 sequence products on the matrix cores (k_mmf16_seq)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_seq_f16.py tests/test_gpu_mega.py -m gpu -q -x -p no:cacheprovider -k "seq or f16 or matches_oracle" 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
export RWKV_BENCH_NO_COLD=1
for v in valu mfma; do
  RWKV_MI_SEQ_F16=$v timeout 400 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_$v.json 2> $O/prefill_7v_$v.err
  python -c "import json; d=json.loads(open('$O/prefill_7v_$v.json').read().strip().splitlines()[-1]); print('prefill rwkv7-2b9 Q5_1 arm $v:', round(d['value'],1), 'tok/s', d.get('parity'))"
done
rm -f /tmp/synthetic-rwkv7*
export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
one() {  # label, lib dir, config, dtype, extra env...
  local label=$1 lib=$2 c=$3 dt=$4; shift 4
  env RWKV_LIB_DIR=$lib "$@" timeout 300 python bench.py --config $c --dtype $dt --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$label $c $dt', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us', flush=True)"
}
for rep in 1 2; do
  one base lib_base rwkv6-7b Q4_0
  one look1 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=1
  one look2 lib rwkv6-7b Q4_0
  one look3 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=3
  one look6 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=6
  one look3_h24 lib rwkv6-7b Q4_0 RWKV_MI_RING_LOOK=3 RWKV_MI_RING_HTHIN=24
done 2>&1 | tee $O/ab.txt
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -44 $O/ring_phase_trace_7b.txt
