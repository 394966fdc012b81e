#!/bin/bash
# kernel + copy timeline of a 4-stage chain of the 1.6B on device 0: what a hop is made of. usage (inside gpurun): tools/gpu_hop_trace.sh <tag>
cd "$(dirname "$0")/.."; T=${1:-r06ht}; O=$PWD/gpurun_out/$T; mkdir -p $O; R=$PWD
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
python bench.py --config rwkv6-1b6 --steps 4 --warmup 1 --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs > /dev/null 2>&1   # (writes the model file)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python $R/bench.py --gpus 4 --chain --chain-devices 0,0,0,0 --config rwkv6-1b6 --steps 24 --warmup 4 --cpu-seconds 0 --parity-tokens 0 > $O/trace.out 2> $O/trace.err
ls -R $O/trace | head
python $R/tools/hop_trace_report.py $O/trace | tee $O/hop_trace.txt
