#!/bin/bash
# Evidence run of round 6: the WHOLE -m gpu suite first (evidence is recorded only when it is green), then the HBM-traffic counter passes of every
# persistent kernel on THIS build (stamped with its sources: bench.py refuses a quote from another build), then tools/gpu_final.sh (bench lines of
# every BASELINE configuration with parity inside, rocprofv3 kernel stats, traces), then the matrix-pipe counters of the sequence GEMM.
# usage: tools/gpu_final_r6.sh <tag>; afterwards tools/collect_profiles.sh <tag>
set -u
cd "$(dirname "$0")/.."
T=${1:-r06f}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
SECONDS=0
( timeout 1500 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider --timeout=400 -rs 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt; echo "suite wall: ${SECONDS}s" >> $O/pytest.txt
if ! grep -q " passed" $O/pytest.txt || grep -q "failed\|error" $O/pytest.txt; then echo "SUITE NOT GREEN: no evidence recorded"; exit 1; fi
R=$PWD
KS=$(python -c "import bench; print(bench.kernel_source_stamp(2))")
for c in "rwkv6-7b 7b" "rwkv6-1b6 1b6"; do cfg=${c% *}; n=${c#* }
  ( cd /tmp
    for ctr in FETCH_SIZE WRITE_SIZE; do
      RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/pmc_${n}_$ctr -o p -- python $R/bench.py --config $cfg --dtype Q4_0 --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/pmc_${n}_$ctr.err
    done )
  python tools/pmc_summary.py $O/pmc_${n}_FETCH_SIZE $O/pmc_${n}_WRITE_SIZE k6_ring $cfg:Q4_0:path2:kind2 profiles/pmc_traffic.json $KS | tee $O/pmc_${n}_summary.txt
done
# (tools/collect_profiles.sh reads pmc_fetch / pmc_write)
ln -sfn pmc_7b_FETCH_SIZE $O/pmc_fetch; ln -sfn pmc_7b_WRITE_SIZE $O/pmc_write
bash tools/gpu_pmc_p47.sh $T/pmc47 > $O/pmc47.log 2>&1; tail -2 $O/pmc47.log
cp profiles/pmc_traffic.json $O/pmc_traffic.json
RWKV_FINAL_SKIP_SUITE=1 RWKV_FINAL_SKIP_PMC=1 bash tools/gpu_final.sh $T
SECONDS=0; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench.py (no flags) wall: ${SECONDS}s" | tee $O/bench_default_wall.txt
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_1b6 -o decode -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_1b6.err )
timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 11 > $O/p47_phase_trace_v4_169m.txt 2>&1
timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 9 > $O/p47_phase_trace_v7_2b9.txt 2>&1
timeout 300 python tools/abi_pinned.py > $O/abi_pinned.txt 2>&1
# matrix-pipe counters of the sequence GEMM (the default arm: k_mmq_mfma) on THIS build, stamped with the sources they ran on
bash tools/gpu_pmc_mfma.sh $T > $O/pmc_mfma.log 2>&1
STAMP=$(python -c "import bench; print(bench.prefill_source_stamp())")
python tools/pmc_mfma_summary.py $O rwkv6-1b6:Q4_0:prefill $STAMP $O/pmc_mfma.json k_mmq_mfma > $O/pmc_mfma_summary.txt 2>&1; cat $O/pmc_mfma_summary.txt
# the one-process chain's hop on this box: 1 / 2 / 4 / 8 stages of the 1.6B and the 7B on device 0, three forms of the hop (tools/gpu_hop_ab.sh)
RWKV_HOP_SKIP_TESTS=1 bash tools/gpu_hop_ab.sh $T/hop > $O/hop.log 2>&1; cp $O/hop/hop_ab.txt $O/hop_ab.txt 2>/dev/null; cat $O/hop_ab.txt
bash tools/gpu_prefill_fuse_ab.sh $T/fuse > $O/fuse.log 2>&1; tail -7 $O/fuse.log > $O/prefill_fuse_ab.txt; cat $O/prefill_fuse_ab.txt
