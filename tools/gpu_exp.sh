#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export RWKV_BENCH_DIR=/tmp
run() { echo "== $*"; env "$@" RWKV_BENCH_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config rwkv6-1b6 --steps 16 --warmup 4 2>&1 | grep -E "Memory access|^\{|Traceback|Error" | cut -c1-130 | head -3; }
run A=1; run A=2; run RWKV_MI_PERSIST=ring RWKV_MI_RING_NO_HEAD=1; run A=3; run RWKV_MI_PERSIST=ring RWKV_MI_RING_NO_HEAD=1
timeout 300 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_api_semantics.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -2
