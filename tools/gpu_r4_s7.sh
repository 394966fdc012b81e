#!/bin/bash
# round 4, step 7: the whole GPU suite (streamed rwkv_eval, F16 sequence arm, ring look 1) + the headline line with the ABI rate + RWKV-7 prefill
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 ) > $O/pytest.txt; cat $O/pytest.txt
export RWKV_BENCH_NO_COLD=1
timeout 500 python bench.py --steps 128 --warmup 16 --cpu-seconds 6 > $O/bench_7b_q4_0.json 2> $O/bench_7b_q4_0.err
python -c "import json; d=json.loads(open('$O/bench_7b_q4_0.json').read().strip().splitlines()[-1]); print('7B Q4_0', round(d['value'],1), 'tok/s  abi', d.get('abi'), 'parity', d.get('parity',{}).get('equal'), 'roof', round(d['roofline']['frac'],4))"
RWKV_MI_ABI_STREAM=0 timeout 300 python bench.py --steps 32 --warmup 8 --cpu-seconds 0 --no-profile > $O/bench_7b_abi_serial.json 2> /dev/null
python -c "import json; d=json.loads(open('$O/bench_7b_abi_serial.json').read().strip().splitlines()[-1]); print('7B Q4_0 serial ABI', d.get('abi'))"
rm -f /tmp/synthetic-rwkv6-7b*
for v in valu mfma; do
  RWKV_MI_SEQ_F16=$v timeout 400 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_$v.json 2> $O/prefill_7v_$v.err
  python -c "import json; d=json.loads(open('$O/prefill_7v_$v.json').read().strip().splitlines()[-1]); print('prefill rwkv7-2b9 Q5_1 arm $v:', round(d['value'],1), 'tok/s', d.get('parity'))"
done
timeout 300 python bench.py --config rwkv7-2b9 --dtype Q5_1 --steps 64 --cpu-seconds 0 > $O/bench_7v.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/bench_7v.json').read().strip().splitlines()[-1]); print('rwkv7-2b9 decode', round(d['value'],1), 'abi', d.get('abi'))"
