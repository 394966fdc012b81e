#!/bin/bash
# sequence-mode GEMM: parity tests of the MFMA path, then its timing on the projection shapes, then a prefill bench line
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
mkdir -p gpurun_out/gemm
timeout 600 python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_prefill.py -x -q -m gpu 2>&1 | tail -5
for f in Q4_0 ${GEMM_FMTS:-}; do echo "== $f"; timeout 200 python tools/gemm_bench.py $f 2>&1 | grep -v "^$" | tail -8; done
timeout 300 python bench.py --mode prefill --config rwkv6-1b6 --dtype Q4_0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/gemm/prefill_1b6_q4_0.json
python -c "
import json; d=json.load(open('gpurun_out/gemm/prefill_1b6_q4_0.json')); r=d['roofline']
print("prefill", round(d["value"],1), "tok/s; gemm", round(r["achieved"],1), "TOP/s avg", round(r["avg_launch_us"],1), "us; parity", d.get("parity"))"
