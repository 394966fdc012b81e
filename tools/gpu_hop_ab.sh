#!/bin/bash
# Round 6: the one-process chain's hop, A/B on one box. usage (inside gpurun): tools/gpu_hop_ab.sh <tag>
# Tests of the chain first; then 1 / 2 / 8 stages of the 1.6B and 1 / 2 / 4 / 8 stages of the 7B on device 0 with the new hop
# (the stage's last layer stores the residual stream in the next stage's buffer, in-launch history, direct launch per stage; RWKV_MI_HOP=copy: one peer copy per hop) against round 5's (RWKV_MI_HOP=mailbox
# RWKV_MI_STAGE_GRAPH=1).
cd "$(dirname "$0")/.."; T=${1:-r06h}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
[ -n "${RWKV_HOP_SKIP_TESTS:-}" ] || ( timeout 900 python -m pytest tests/test_gpu_pipeline_cpp.py tests/test_gpu_pipeline.py tests/test_gpu_ipc_ranks.py tests/test_gpu_mega.py tests/test_gpu_persist_v47.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) | tee $O/pytest_pipeline.txt
B="timeout 400 python bench.py --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d['value'],1), 'tokens/s', round(d['ms_per_step'],4), 'ms', (d.get('parity') or {}).get('equal'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in rwkv6-1b6 rwkv6-7b; do
  $B --config $cfg --steps 128 --warmup 16 > $O/one_$cfg.json 2> $O/one_$cfg.err; line $O/one_$cfg.json
  for devs in 0,0 0,0,0,0 0,0,0,0,0,0,0,0; do
    n=$(echo $devs | tr ',' '\n' | wc -l)
    $B --gpus $n --chain --chain-devices $devs --config $cfg --steps 128 --warmup 16 > $O/chain${n}_$cfg.json 2> $O/chain${n}_$cfg.err; line $O/chain${n}_$cfg.json
    RWKV_MI_HOP=copy $B --gpus $n --chain --chain-devices $devs --config $cfg --steps 128 --warmup 16 > $O/chain${n}_$cfg.copy.json 2> $O/chain${n}_$cfg.copy.err; line $O/chain${n}_$cfg.copy.json
    RWKV_MI_HOP=mailbox RWKV_MI_STAGE_GRAPH=1 $B --gpus $n --chain --chain-devices $devs --config $cfg --steps 128 --warmup 16 > $O/chain${n}_$cfg.r5.json 2> $O/chain${n}_$cfg.r5.err; line $O/chain${n}_$cfg.r5.json
  done
done | tee $O/hop_ab.txt
