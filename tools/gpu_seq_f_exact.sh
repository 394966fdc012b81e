#!/bin/bash
# Round 6: the exact F16 / F32 sequence arm on the matrix cores (k_mmfx_seq): the instruction's addition order, the tests, and sequence passes
# of an FP16 / FP32 file and of the RWKV-7 2.9B (F16 low-rank stages) against RWKV_MI_SEQ_F=valu (k_mvf token tiles), same box.
cd "$(dirname "$0")/.."; T=${1:-r06x}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
tools/mfma_f32_chain | tee $O/mfma_f32_chain.txt
( timeout 900 python -m pytest tests/test_gpu_seq_f_exact.py tests/test_gpu_seq_f16.py tests/test_gpu_mul_mat.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6 ) | tee $O/pytest.txt
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
for c in "rwkv7-2b9 Q5_1 3 128" "rwkv6-1b6 FP16 3 64" "rwkv6-1b6 FP32 3 64"; do set -- $c
  timeout 600 python bench.py --mode prefill --config $1 --dtype $2 --steps $3 --warmup 1 --cpu-seconds 12 --parity-tokens $4 > $O/prefill_$1_$2.json 2> $O/prefill_$1_$2.err; line $O/prefill_$1_$2.json
  RWKV_MI_SEQ_F=valu timeout 600 python bench.py --mode prefill --config $1 --dtype $2 --steps $3 --warmup 1 --cpu-seconds 0 > $O/prefill_$1_$2.valu.json 2> $O/prefill_$1_$2.valu.err; line $O/prefill_$1_$2.valu.json
done 2>&1 | tee $O/seq_f_exact.txt
