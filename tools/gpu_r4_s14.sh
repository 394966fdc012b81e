#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04o; mkdir -p $O
export TMPDIR=/tmp
for L in lib_prev lib lib_w7b; do
  ( RWKV_LIB_DIR=$L timeout 100 python -m pytest tests/test_gpu_prefill.py -m gpu -v -x -p no:cacheprovider -o faulthandler_timeout=60 -k "wkv7_sequence" 2>&1 | tail -40 ) > $O/edges_$L.txt 2>&1
  echo "== $L"; tail -15 $O/edges_$L.txt
done
