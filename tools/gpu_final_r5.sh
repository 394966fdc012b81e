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
RWKV_FINAL_SKIP_SUITE=1 bash tools/gpu_final.sh $T
timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 11 > $O/p47_phase_trace_v4_169m.txt 2>&1
timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 9 > $O/p47_phase_trace_v7_2b9.txt 2>&1
# matrix-pipe counters of the sequence GEMM on THIS build, stamped with the sources they ran on
bash tools/gpu_pmc_mfma.sh $T > $O/pmc_mfma.log 2>&1
STAMP=$(python -c "import bench; print(bench.prefill_source_stamp())")
python tools/pmc_mfma_summary.py $O rwkv6-1b6:Q4_0:prefill $STAMP $O/pmc_mfma.json > $O/pmc_mfma_summary.txt 2>&1; cat $O/pmc_mfma_summary.txt
