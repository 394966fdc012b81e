#!/bin/bash
# quick look at the ring kernel on one box: bit-exactness (two geometries), ring vs regs on the 7B and the 1.6B, phase trace, knob sweep
# usage: tools/gpu_ring_quick.sh <outdir-tag> ["ENV=.. ENV=.." knob settings to sweep, one string per argument]
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03x}; mkdir -p $O; shift
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_MI_NO_AUTOTUNE=1
ok=1
for a in "mega-v6-2048 Q4_0 direct" "mega-v6-4096 Q4_0 direct" "mega-v6-4096-v4k Q4_0 direct" "mega-v6-2048-v8k Q5_1 direct" "mega-v6-2048-v32k Q4_0 direct" "mega-v6-4096 Q8_0 direct"; do
  f=$O/dbg_$(echo $a | tr ' ' '_').txt
  RWKV_MI_PERSIST=ring timeout 150 python tools/dbg_fused.py $a > $f 2>&1; echo "rc $? $(grep -E 'RESULT|path' $f | tr '\n' ' ')"
  grep -q "OK" $f || { ok=0; tail -20 $f; }
done
[ $ok = 1 ] || exit 1
b() { timeout 300 python bench.py --config $2 --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 --parity-tokens ${3:-0} 2>$O/bench.err | tee $O/bench_$1_$2.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$1 $2', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us; frac', round(r.get('frac',0),4), 'parity', d.get('parity',{}).get('equal'))"; }
RWKV_MI_PERSIST=regs b regs rwkv6-7b
RWKV_MI_PERSIST=ring b ring rwkv6-7b 64
RWKV_MI_PERSIST=regs b regs rwkv6-1b6
RWKV_MI_PERSIST=ring b ring rwkv6-1b6 64
timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_cycles_7b.txt 2> $O/trace_7b.err; cat $O/ring_phase_cycles_7b.txt
timeout 200 python tools/trace_ring.py rwkv6-1b6 5 > $O/ring_phase_cycles_1b6.txt 2> $O/trace_1b6.err
for k in "$@"; do
  env $k RWKV_MI_PERSIST=ring timeout 300 python bench.py --config rwkv6-7b --dtype Q4_0 --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 --parity-tokens 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$k', round(d['value'],1), 'tok/s; kernel', round(r.get('avg_launch_us',0),1), 'us')"
done
