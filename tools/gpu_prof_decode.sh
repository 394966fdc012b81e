#!/bin/bash
# usage: gpu_prof_decode.sh <outdir> <config> <dtype> [env...]
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o decode -- python $R/bench.py --config $2 --dtype $3 --steps 48 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > $R/$O/bench_prof.json 2> $R/$O/rocprof.err
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -${4:-12} "$f" | cut -c1-150
