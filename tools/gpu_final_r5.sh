#!/bin/bash
# Evidence run of round 5: the WHOLE -m gpu suite first; evidence is recorded only when it finished green (round 4 ended with its last
# suite run killed by its own timeout behind the experiments). usage: tools/gpu_final_r5.sh <tag>
set -u
cd "$(dirname "$0")/.."
T=${1:-r05z}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
SECONDS=0
( timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt; echo "suite wall: ${SECONDS}s" >> $O/pytest.txt
if ! grep -q " passed" $O/pytest.txt || grep -q "failed\|error" $O/pytest.txt; then echo "SUITE NOT GREEN: no evidence recorded"; exit 1; fi
# HBM traffic of the decode kernel FIRST (separate FETCH_SIZE / WRITE_SIZE passes), summarised into profiles/pmc_traffic.json under the hash of
# this build's sources: the bench lines below then quote counters of the build they run on
R=$PWD
( cd /tmp
  RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/pmc_fetch.err
  RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/pmc_write.err )
KS=$(python -c "import bench; print(bench.kernel_source_stamp(2))")
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write k6_ring rwkv6-7b:Q4_0:path2:kind2 profiles/pmc_traffic.json $KS > $O/pmc_traffic_summary.txt 2>&1; cat $O/pmc_traffic_summary.txt; cp profiles/pmc_traffic.json $O/pmc_traffic.json
RWKV_FINAL_SKIP_SUITE=1 RWKV_FINAL_SKIP_PMC=1 bash tools/gpu_final.sh $T
timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 11 > $O/p47_phase_trace_v4_169m.txt 2>&1
timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 9 > $O/p47_phase_trace_v7_2b9.txt 2>&1
# matrix-pipe counters of the sequence GEMM on THIS build, stamped with the sources they ran on (skipped when the committed quote carries this build's hash)
if python -c "import json,bench,sys; sys.exit(0 if json.load(open('profiles/pmc_mfma.json'))['rwkv6-1b6:Q4_0:prefill'].get('prefill_source_stamp')==bench.prefill_source_stamp() else 1)"; then echo 'pmc_mfma.json is of this build'; exit 0; fi
bash tools/gpu_pmc_mfma.sh $T > $O/pmc_mfma.log 2>&1
STAMP=$(python -c "import bench; print(bench.prefill_source_stamp())")
python tools/pmc_mfma_summary.py $O rwkv6-1b6:Q4_0:prefill $STAMP $O/pmc_mfma.json > $O/pmc_mfma_summary.txt 2>&1; cat $O/pmc_mfma_summary.txt
