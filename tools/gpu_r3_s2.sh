#!/bin/bash
# Round-3 session 2: first run of the LDS-DMA ring kernel (ring_v6.hip): bit-exactness on the two geometries, then same-box A/B against the
# register-prefetch kernel on the 7B and the 1.6B, the phase trace, and the persistent-kernel / stage-chain tests.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1
for a in "mega-v6-2048 Q4_0 direct" "mega-v6-4096 Q4_0 direct" "mega-v6-4096 Q5_1 direct" "mega-v6-2048 Q8_0 direct" "mega-v6-4096 Q4_1 direct" "mega-v6-2048 Q5_0 direct"; do
  RWKV_MI_PERSIST=ring timeout 150 python tools/dbg_fused.py $a > $O/dbg_$(echo $a | tr ' ' '_').txt 2>&1; echo "rc $? $(grep -E 'RESULT|path' $O/dbg_$(echo $a | tr ' ' '_').txt | tr '\n' ' ')"
done
if grep -q "RESULT mega-v6-4096 Q4_0 direct OK" $O/dbg_mega-v6-4096_Q4_0_direct.txt; then
  for rep in 1 2; do for v in regs ring; do for c in rwkv6-7b rwkv6-1b6; do
    RWKV_MI_PERSIST=$v timeout 300 python bench.py --config $c --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 2>$O/bench_${v}_${c}.err | tee $O/bench_${v}_${c}_$rep.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$v $c', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us; parity', d.get('parity',{}).get('equal'))"
  done; done; done
  timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_cycles_7b.txt 2> $O/trace_7b.err; cat $O/ring_phase_cycles_7b.txt
  timeout 200 python tools/trace_ring.py rwkv6-1b6 5 > $O/ring_phase_cycles_1b6.txt 2> $O/trace_1b6.err
  for k in "RWKV_MI_RING_INFLIGHT=32" "RWKV_MI_RING_INFLIGHT=56" "RWKV_MI_RING_THIN=48" "RWKV_MI_RING_THIN=8" "RWKV_MI_RING_KB=64" "RWKV_MI_RING_KB=80"; do
    env $k RWKV_MI_PERSIST=ring timeout 300 python bench.py --config rwkv6-7b --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$k', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us')"
  done
fi
( timeout 600 python -m pytest tests/test_gpu_mega.py tests/test_gpu_pipeline_cpp.py tests/test_gpu_pipeline.py -q -x -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.txt; tail -3 $O/pytest.txt
