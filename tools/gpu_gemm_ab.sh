#!/bin/bash
# parity tests of the current GEMM, then same-box timing: lib_prev (reference build) vs lib, alternating
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
timeout 600 python -m pytest tests/test_gpu_mul_mat.py tests/test_gpu_prefill.py -x -q -m gpu 2>&1 | tail -3
for v in lib_prev lib lib_prev lib; do echo "== $v"; RWKV_LIB_DIR=$v timeout 200 python tools/gemm_bench.py ${FMT:-Q4_0} 2>&1 | grep time_mm | head -${NSHAPES:-3}; done
