#!/bin/bash
# Round-3 session 1: (1) the LDS-DMA ring microbenchmark of DESIGN.md section 9.1 (never run before), (2) the stress loop VERDICT r02 asked
# for: the sequence-pass and stage-chain tests repeated on ONE box (one unexplained failure + one unexplained hang).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 300 tools/ring_bench 4 3 ) > $O/ring_bench.txt 2>&1; tail -12 $O/ring_bench.txt
: > $O/stress.txt
for i in $(seq 1 ${1:-30}); do
  s=$(date +%s%N)
  timeout 240 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_pipeline_cpp.py -q -x -m gpu -k "sequence_pass or stage_chain" -p no:cacheprovider > $O/stress_$i.log 2>&1
  rc=$?
  e=$(date +%s%N)
  echo "iter $i rc $rc ms $(( (e - s) / 1000000 )) $(tail -1 $O/stress_$i.log)" | tee -a $O/stress.txt
  if [ $rc -ne 0 ]; then cp $O/stress_$i.log $O/stress_FAIL_$i.log; fi
  rm -f $O/stress_$i.log
done
