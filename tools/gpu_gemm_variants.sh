#!/bin/bash
# timing-only variants of the GEMM (debug builds lib_d*): which wait costs what
cd "$(dirname "$0")/.."
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
for v in lib ${VARIANTS:-lib_d1 lib_d2 lib_d3}; do echo "== $v"; RWKV_LIB_DIR=$v timeout 200 python tools/gemm_bench.py Q4_0 2>&1 | grep time_mm | head -3; done
