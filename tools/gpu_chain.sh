#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline_cpp.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python bench.py --gpus 2 --chain --chain-devices 0,0 --config rwkv6-1b6 --steps 64 --warmup 8 2>&1 | tail -2
timeout 300 python bench.py --gpus 1 --chain --config rwkv6-1b6 --steps 64 --warmup 8 2>&1 | tail -1
