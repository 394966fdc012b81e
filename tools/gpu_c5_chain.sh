#!/bin/bash
# BASELINE configuration C5's shape on ONE GPU: RWKV-6 7B Q8_0 through 1 / 2 / 4 / 8 pipeline stages (every stage on device 0, the one-process
# chain of RWKV_MI_DEVICES; the RCCL form needs one GPU per rank). Single-stream tokens/s, the aggregate of N streams in flight, parity of the
# chain against a one-device context inside each run.
cd "$(dirname "$0")/.."; T=${1:-r06c5}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
B="timeout 600 python bench.py --config rwkv6-7b --dtype Q8_0 --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs --steps 96 --warmup 8"
$B > $O/one.json 2> $O/one.err
for devs in 0,0 0,0,0,0 0,0,0,0,0,0,0,0; do n=$(echo $devs | tr ',' '\n' | wc -l); $B --gpus $n --chain --chain-devices $devs > $O/chain$n.json 2> $O/chain$n.err; done
python - $O <<'PY' | tee $O/c5_chain.txt
import json,sys
O=sys.argv[1]
for f in ("one","chain2","chain4","chain8"):
    try:
        d=json.loads(open(f"{O}/{f}.json").read().strip().splitlines()[-1]); m=d.get("multi_stream") or {}
        print(f"{f:7s} single stream {d['value']:7.1f} tokens/s {d['ms_per_step']:.4f} ms   {m.get('streams','-')} streams in flight {m.get('tokens_per_s_aggregate',0):7.1f} tokens/s   parity {(d.get('parity') or {}).get('equal')}")
    except Exception as e:
        print(f, "FAILED", e)
PY
