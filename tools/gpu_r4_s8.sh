#!/bin/bash
# round 4, step 8: whole suite (no -x) + F16 sequence arm on RWKV-7 2.9B and on an FP16 file
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 ) > $O/pytest.txt; cat $O/pytest.txt
export RWKV_BENCH_NO_COLD=1
for v in valu mfma; do
  RWKV_MI_SEQ_F16=$v timeout 400 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_$v.json 2> $O/prefill_7v_$v.err
  python -c "import json; d=json.loads(open('$O/prefill_7v_$v.json').read().strip().splitlines()[-1]); print('prefill rwkv7-2b9 Q5_1 arm $v:', round(d['value'],1), 'tok/s', d.get('parity'))"
done
rm -f /tmp/synthetic-rwkv7*
for v in valu mfma; do
  RWKV_MI_SEQ_F16=$v timeout 400 python bench.py --config rwkv6-1b6 --dtype FP16 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > $O/prefill_1b6_fp16_$v.json 2> $O/prefill_1b6_fp16_$v.err
  python -c "import json; d=json.loads(open('$O/prefill_1b6_fp16_$v.json').read().strip().splitlines()[-1]); print('prefill rwkv6-1b6 FP16 arm $v:', round(d['value'],1), 'tok/s')"
done
