#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r02h}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
timeout 300 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 4 > $O/prefill.json 2> $O/prefill.err
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/$O/pmc_mfma -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA --output-format csv -d $R/$O/pmc_mfma2 -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma2.err
cd $R
tail -c 900 $O/prefill.json; tail -3 $O/pmc_mfma.err; tail -3 $O/pmc_mfma2.err
find $O -name "*counter_collection.csv" | head
