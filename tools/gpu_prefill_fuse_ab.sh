#!/bin/bash
# Round 6: sequence mode with the quantiser folded into its producers (key-product epilogue, group norm, first RWKV-6 mix), A/B on one box.
cd "$(dirname "$0")/.."; T=${1:-r06p}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
[ -n "${RWKV_HOP_SKIP_TESTS:-}" ] || ( timeout 1200 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_real_geometry.py tests/test_gpu_tiny_rwkv.py tests/test_gpu_synthetic.py tests/test_gpu_reference_programs.py tests/test_gpu_prefill_fast.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) | tee $O/pytest_seq.txt
B="timeout 500 python bench.py --mode prefill --config rwkv6-1b6 --dtype Q4_0 --steps 6 --warmup 2"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f=d.get('fast_arms',{})
    print(sys.argv[1].split('/')[-1], 'default', round(d['value']), 'tokens/s', round(d['ms_per_step'],3), 'ms; opt-in arms', round(f.get('tokens_per_s',0)), round(f.get('ms_per_step',0),3), 'parity', (d.get('parity') or {}).get('equal'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
$B > $O/prefill_new.json 2> $O/prefill_new.err; line $O/prefill_new.json
RWKV_MI_NO_EPI_QUANT=1 RWKV_MI_NO_GN_QUANT=1 RWKV_MI_NO_MIX_QUANT=1 $B --cpu-seconds 0 > $O/prefill_old.json 2> $O/prefill_old.err; line $O/prefill_old.json
$B --cpu-seconds 0 > $O/prefill_new2.json 2> $O/prefill_new2.err; line $O/prefill_new2.json
RWKV_MI_NO_EPI_QUANT=1 $B --cpu-seconds 0 > $O/prefill_noepi.json 2> $O/prefill_noepi.err; line $O/prefill_noepi.json
RWKV_MI_NO_GN_QUANT=1 $B --cpu-seconds 0 > $O/prefill_nogn.json 2> $O/prefill_nogn.err; line $O/prefill_nogn.json
timeout 500 python bench.py --mode prefill --config rwkv7-2b9 --dtype Q5_1 --steps 3 --warmup 1 > $O/prefill_v7.json 2> $O/prefill_v7.err; line $O/prefill_v7.json
