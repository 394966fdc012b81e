#!/bin/bash
# Round-5 GPU sessions, one parameterised script (replaces the per-session gpu_r4_s*.sh files): tools/gpu_r5.sh <step> [args]
#   p47        the persistent RWKV-4 / RWKV-7 launch: its tests, then the C2 / C4 decode lines with parity
#   p47bench   only the two decode lines (+ kernel stats)
#   suite      the whole -m gpu suite (evidence: gpurun_out/r05/pytest.txt), refuses to continue when it did not finish
set -u
cd "$(dirname "$0")/.."
STEP=${1:-p47}; shift || true
O=gpurun_out/r05_$STEP; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
line() { python -c "
import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); r=d.get('roofline',{}); print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms', 'frac', r.get('frac'), 'parity', d.get('parity'), 'abi', d.get('abi'), 'path', d.get('config',{}).get('decode_path'), flush=True)
"; }
bench_one() {   # name config dtype extra...
  local n=$1 c=$2 t=$3; shift 3
  timeout 300 python bench.py --config $c --dtype $t --steps 256 --warmup 16 --cpu-seconds 0 "$@" > $O/bench_$n.json 2> $O/bench_$n.err; tail -1 $O/bench_$n.json | line $n
}
case $STEP in
p47all)
  bash $0 p47 "$@"; bash $0 p47trace
  ;;
p47)
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_persist_v47.py -m gpu -v -x -p no:cacheprovider "$@" > $O/pytest_p47_full.txt 2>&1; grep -v "^  File\|^$" $O/pytest_p47_full.txt | head -60 > $O/pytest_p47.txt; tail -5 $O/pytest_p47_full.txt >> $O/pytest_p47.txt; cat $O/pytest_p47.txt
  bench_one v4 rwkv4-169m Q5_1
  bench_one v7 rwkv7-2b9 Q5_1
  ;;
p47bench)
  bench_one v4 rwkv4-169m Q5_1 "$@"
  bench_one v7 rwkv7-2b9 Q5_1 "$@"
  ;;
p47trace)
  timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 11 > $O/trace_v4.txt 2>&1; cat $O/trace_v4.txt | tail -40
  timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 9 > $O/trace_v7.txt 2>&1; cat $O/trace_v7.txt | tail -45
  R=$PWD
  for c in "v4 rwkv4-169m" "v7 rwkv7-2b9"; do n=${c% *}; cfg=${c#* }
    ( cd /tmp && rm -rf /tmp/prof_$n && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o p -- python $R/bench.py --config $cfg --dtype Q5_1 --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /tmp/prof_$n.log 2>&1 ) || tail -3 /tmp/prof_$n.log
    f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp $f $O/decode_${n}_kernel_stats.csv; head -8 $f | cut -c1-160; fi
  done
  ;;
calm)   # long waits on one unit (RWKV_MI_P47_CALM bit 0: tail, bit 1: head workgroups, bit 2: row polling waves): A/B on one box, raw stamps
  for c in ${CALMS:-1 5 7 1 5}; do
    export RWKV_MI_P47_CALM=$c
    bench_one v4_calm$c rwkv4-169m Q5_1 --abi-tokens 0 --parity-tokens 16
    bench_one v7_calm$c rwkv7-2b9 Q5_1 --abi-tokens 0 --parity-tokens 16
  done
  for c in ${TCALMS:-1 5}; do
    export RWKV_MI_P47_CALM=$c
    for ly in 3 11; do TRACE_DUMP=$O/raw_v4_calm${c}_l$ly.npy timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 $ly > $O/trace_v4_calm${c}_l$ly.txt 2>&1; done
    for ly in 9 20; do TRACE_DUMP=$O/raw_v7_calm${c}_l$ly.npy timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 $ly > $O/trace_v7_calm${c}_l$ly.txt 2>&1; done
  done
  grep -h "layer wall" $O/trace_*_l*.txt
  ;;
ab)     # A/B/A/B of lib/ against a variant build of persist_v47.hip in lib_b/ (tools/build_variant.sh lib_b -D...): parity first
  timeout 900 python -X faulthandler -m pytest ${ABTESTS:-tests/test_gpu_persist_v47.py} -m gpu -x -q -p no:cacheprovider > $O/pytest_p47.txt 2>&1; tail -3 $O/pytest_p47.txt
  L=rwkv.cpp_amd/lib/librwkv.so; cp $L /tmp/lib_main.so
  for v in main b main b; do
    if [ $v = b ]; then cp rwkv.cpp_amd/lib_b/librwkv.so $L; else cp /tmp/lib_main.so $L; fi
    for c in ${CONFIGS:-v4:rwkv4-169m:Q5_1 v7:rwkv7-2b9:Q5_1}; do IFS=: read n cfg dt <<< "$c"; bench_one ${n}_$v $cfg ${dt:-Q5_1} --abi-tokens 0; done
  done
  cp /tmp/lib_main.so $L
  if [ -z "${NO_TRACE:-}" ]; then
  for ly in 3 11; do timeout 200 python tools/trace_p47.py rwkv4-169m Q5_1 $ly > $O/trace_v4_l$ly.txt 2>&1; done
  grep -h "layer wall" $O/trace_*_l*.txt; sed -n 2,26p $O/trace_v4_l11.txt
  fi
  if [ -n "${TRACE_V7:-}" ]; then timeout 300 python tools/trace_p47.py rwkv7-2b9 Q5_1 9 > $O/trace_v7_l9.txt 2>&1; sed -n 2,31p $O/trace_v7_l9.txt; fi
  ;;
pffdbg)
  for d in 0 1 2 3 4 7; do RWKV_MI_PFF_DBG=$d timeout 200 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 4 --warmup 1 --cpu-seconds 0 --parity-tokens 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('dbg $d', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms', 'gemm avg us', round(r['avg_launch_us'],2), flush=True)"; done
  ;;
prefillbench)
  for a in fast exact; do RWKV_MI_SEQ_Q=$a timeout 300 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 0 --parity-tokens 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$a', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms', 'gemm avg us', round(r['avg_launch_us'],2), 'TOP/s', round(r['achieved'],1), flush=True)"; done
  ;;
prefill)
  timeout ${PYT:-240} python -m pytest tests/test_gpu_prefill_fast.py -m gpu -q -x -p no:cacheprovider --durations=5 "$@" > $O/pytest_fast_full.txt 2>&1; tail -15 $O/pytest_fast_full.txt | grep -v "^$" > $O/pytest_fast.txt; cat $O/pytest_fast.txt
  pline() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); p=d.get('parity',{})
print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms/pass', 'gemm TOP/s', round(r.get('achieved',0),1), 'avg_us', round(r.get('avg_launch_us',0),2), 'launches', r.get('launches'), 'parity', p.get('equal'), p.get('exact_arm_bit_identical'), json.dumps(p.get('timed_arm',{}))[:400], flush=True)
"; }
  timeout 400 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 25 --parity-tokens ${PARITY:-128} > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err; tail -1 $O/prefill_1b6_q4_0.json | pline 1b6
  RWKV_MI_SEQ_Q=exact timeout 400 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 0 --parity-tokens 0 > $O/prefill_1b6_q4_0_exact.json 2>> $O/prefill_1b6.err; tail -1 $O/prefill_1b6_q4_0_exact.json | pline 1b6-exact
  timeout 400 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err; tail -1 $O/prefill_7v_2b9_q5_1.json | pline 2b9
  ;;
default)
  # the driver's own command: headline + other_configs, timed
  SECONDS=0; timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench.py (no flags) wall: ${SECONDS}s"; tail -1 $O/bench_default.json | line default
  python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
for k,v in d.get("other_configs",{}).items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("tokens_per_s","kernel_frac_of_8TBps","token_frac_of_8TBps","parity","error","seconds","persist_kind")})
PY
  tail -3 $O/bench_default.err
  ;;
suite)
  ( timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
  grep -q " passed" $O/pytest.txt || { echo "SUITE DID NOT FINISH: no evidence recorded"; exit 1; }
  ;;
esac
