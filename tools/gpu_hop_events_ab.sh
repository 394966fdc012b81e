cd /root/repo; O=gpurun_out/r06h3; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_pipeline_cpp.py tests/test_gpu_pipeline.py tests/test_gpu_ipc_ranks.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 ) | tee $O/pytest.txt
B="timeout 400 python bench.py --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs --steps 128 --warmup 16"
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); m=d.get('multi_stream') or {}
    print(sys.argv[1].split('/')[-1], round(d['value'],1), 'tokens/s', round(d['ms_per_step'],4), 'ms', 'streams', round(m.get('tokens_per_s_aggregate',0),1), (d.get('parity') or {}).get('equal'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
for cfg in rwkv6-1b6 rwkv6-7b; do
  $B --config $cfg > $O/one_$cfg.json 2>/dev/null; line $O/one_$cfg.json
  for devs in 0,0 0,0,0,0 0,0,0,0,0,0,0,0; do n=$(echo $devs | tr ',' '\n' | wc -l)
    $B --config $cfg --gpus $n --chain --chain-devices $devs > $O/chain${n}_$cfg.json 2>/dev/null; line $O/chain${n}_$cfg.json
    RWKV_MI_HOP_TAKEN=1 RWKV_MI_HOP_OWN_EVENT=1 $B --config $cfg --gpus $n --chain --chain-devices $devs > $O/chain${n}_$cfg.prev.json 2>/dev/null; line $O/chain${n}_$cfg.prev.json
  done
done | tee $O/hop3.txt
