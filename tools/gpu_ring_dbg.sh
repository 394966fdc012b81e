#!/bin/bash
# timing experiments on the ring kernel (RWKV_MI_RING_DBG: the results of these runs are wrong on purpose): where does a record's time go?
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03dbg}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring 
for d in ${2:-0 3 4 7}; do
  echo "=== DBG $d"
  RWKV_MI_RING_DBG=$d timeout 200 python tools/trace_ring.py rwkv6-7b 5 2>$O/err_$d.txt | tee $O/trace_dbg$d.txt | grep -E "rows|prologue|layer wall|LOADER|consumer layer"
done
