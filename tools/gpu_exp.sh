#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp RWKV_MI_PERSIST=ring RWKV_MI_NO_AUTOTUNE=1
for a in "mega-v6-2048 Q4_0" "mega-v6-4096 Q4_0" "mega-v6-4096 Q5_1" "mega-v6-2048-v8k Q8_0" "mega-v6-4096 Q8_0" "mega-v6-4096-v4k Q4_0"; do timeout 120 python tools/dbg_fused.py $a direct 2>&1 | grep RESULT; done
python - <<'PY'
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, oracle_lib as O
from gpu_lib import library, model, synth
library()
for fmt in ("Q4_1", "Q5_0"):
    p = '/tmp/x_%s.bin' % fmt
    synth.write_model(p, synth.CONFIGS["mega-v6-4096"], fmt, seed=7)
    om = O.OracleModel(p); m = model(p); ost, st, ok = om.init_state(), None, True
    for t in [1, 2, 3, 400, 5, 77]:
        ol, ost = om.eval(t, ost); gl, st = m.eval(t, st); ok = ok and np.array_equal(gl, ol) and np.array_equal(st, ost)
    print('RESULT mega-v6-4096', fmt, 'OK' if ok else 'MISMATCH', flush=True)
PY
for rep in 1 2 3 4; do for lib in lib_base lib; do
  RWKV_LIB_DIR=$lib timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('7b $lib', round(d['value'],1), round(r['avg_launch_us'],1))"; done; done
for lib in lib_base lib; do RWKV_LIB_DIR=$lib timeout 300 python bench.py --config rwkv6-7b --dtype Q8_0 --steps 64 --warmup 8 --cpu-seconds 0 --parity-tokens 0 --abi-tokens 0 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('7b Q8_0 $lib', round(d['value'],1))"; done
