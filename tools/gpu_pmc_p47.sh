#!/bin/bash
# HBM traffic of the persistent RWKV-4 / RWKV-7 decode kernel (k47_persist) from the PMC counters: FETCH_SIZE and WRITE_SIZE in separate
# passes over a short decode of each BASELINE file, summarised into profiles/pmc_traffic.json under the hash of this build's sources.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-pmc47}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1 RWKV_BENCH_NO_COLD=1
R=$PWD
KS=$(python -c "import bench; print(bench.kernel_source_stamp(3))")
for c in "rwkv7-2b9 v7" "rwkv4-169m v4"; do cfg=${c% *}; n=${c#* }
  ( cd /tmp
    for ctr in FETCH_SIZE WRITE_SIZE; do
      timeout 200 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/${n}_$ctr -o p -- python $R/bench.py --config $cfg --dtype Q5_1 --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/${n}_$ctr.err
    done )
  python tools/pmc_summary.py $O/${n}_FETCH_SIZE $O/${n}_WRITE_SIZE k47_persist $cfg:Q5_1:path2:kind3 profiles/pmc_traffic.json $KS | tee $O/${n}_summary.txt
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
