#!/bin/bash
# k_mmfx_seq: which instantiation for which shapes, same box. (Ran at commit 'sequence mode: F16 / F32 matrices on the matrix cores ...' + the
# partial split, when RWKV_MI_FX = wide | split | split2 forced one of three instantiations; 'split' won everywhere and is the only one built
# now -- the switch is gone, the record is profiles/r06_seq_f_exact.txt.)
cd "$(dirname "$0")/.."; T=${1:-r06y}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_seq_f_exact.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 ) | tee $O/pytest.txt
for fx in split split2 wide; do ( RWKV_MI_FX=$fx timeout 900 python -m pytest tests/test_gpu_seq_f_exact.py -x -q -m gpu -p no:cacheprovider -k "exact_gemm" 2>&1 | tail -1 ) | tee -a $O/pytest.txt; done
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], 'default arms', round(d['value']), 'tokens/s', round(d['ms_per_step'],3), 'ms')
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for c in "rwkv7-2b9 Q5_1" "rwkv6-1b6 FP16" "rwkv6-1b6 FP32"; do set -- $c
  for fx in default split split2 wide; do
    if [ $fx = default ]; then unset RWKV_MI_FX; else export RWKV_MI_FX=$fx; fi
    timeout 600 python bench.py --mode prefill --config $1 --dtype $2 --steps 3 --warmup 1 --cpu-seconds 0 > $O/prefill_$1_$2.$fx.json 2> $O/prefill_$1_$2.$fx.err; line $O/prefill_$1_$2.$fx.json
  done
done 2>&1 | tee $O/fx_forms.txt
