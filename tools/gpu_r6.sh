#!/bin/bash
# Round-6 GPU sessions: tools/gpu_r6.sh <step> [args]   (outputs under gpurun_out/r06_<step>/)
set -u
cd "$(dirname "$0")/.."
STEP=${1:-diag}; shift || true
O=gpurun_out/r06_$STEP; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp RWKV_BENCH_NO_COLD=1
line() { python -c "
import json,sys
for l in sys.stdin.read().strip().splitlines()[-1:]:
    d=json.loads(l); r=d.get('roofline',{}); print('$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],4), 'ms', 'kernel_us', round(r.get('avg_launch_us',0),1), 'frac', r.get('frac'), 'parity', d.get('parity'), 'path', d.get('config',{}).get('decode_path'), flush=True)
"; }
bench_one() {   # name config dtype extra...
  local n=$1 c=$2 t=$3; shift 3
  timeout 300 python bench.py --config $c --dtype $t --steps ${STEPS:-256} --warmup 16 --cpu-seconds 0 --abi-tokens 0 --no-other-configs "$@" > $O/bench_$n.json 2> $O/bench_$n.err; tail -1 $O/bench_$n.json | line $n
}
case $STEP in
diag)   # where round 5's kernel stands on this box + the watch-path microbenchmark + the tl poll without its watch stage
  ./tools/watch_bench > $O/watch_bench.txt 2>&1; cat $O/watch_bench.txt
  export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
  bench_one base1 rwkv6-7b Q4_0 --parity-tokens 16
  RWKV_MI_RING_DBG=128 bench_one tldirect1 rwkv6-7b Q4_0 --parity-tokens 16
  bench_one base2 rwkv6-7b Q4_0 --parity-tokens 0
  RWKV_MI_RING_DBG=128 bench_one tldirect2 rwkv6-7b Q4_0 --parity-tokens 0
  RWKV_MI_RING_LTRACE=/tmp/ltrace.bin timeout 300 python tools/trace_ring.py rwkv6-7b 5 > $O/trace_7b.txt 2>&1; head -60 $O/trace_7b.txt
  RWKV_MI_RING_DBG=128 timeout 300 python tools/trace_ring.py rwkv6-7b 5 > $O/trace_7b_tldirect.txt 2>&1; head -36 $O/trace_7b_tldirect.txt
  ;;
ab)     # A/B/A/B of lib/ against variant builds in lib_<x>/ (tools/build_variant.sh lib_<x> ring_v6 -D...): VARIANTS="b c", parity first
  L=rwkv.cpp_amd/lib/librwkv.so; cp $L /tmp/lib_main.so
  export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
  for v in ${VARIANTS:-b}; do
    cp rwkv.cpp_amd/lib_$v/librwkv.so $L
    timeout 600 python -X faulthandler -m pytest ${ABTESTS:-tests/test_gpu_mega.py} -m gpu -x -q -p no:cacheprovider > $O/pytest_$v.txt 2>&1; echo "variant $v: $(tail -1 $O/pytest_$v.txt)"
  done
  for rep in 1 2; do for v in main ${VARIANTS:-b}; do
    if [ $v = main ]; then cp /tmp/lib_main.so $L; else cp rwkv.cpp_amd/lib_$v/librwkv.so $L; fi
    for c in ${CONFIGS:-7b:rwkv6-7b:Q4_0}; do IFS=: read n cfg dt <<< "$c"; bench_one ${n}_${v}_$rep $cfg ${dt:-Q4_0} --parity-tokens ${PARITY:-16}; done
  done; done
  cp /tmp/lib_main.so $L
  ;;
fold)   # embedding + ln0 + argmax inside k6_ring: tests, parity against the CPU oracle, A/B against the separate launches (RWKV_MI_RING_NO_EMBED=1) and variants
  export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_mega.py tests/test_gpu_sampling.py tests/test_gpu_abi_stream.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
  timeout 300 python bench.py --config rwkv6-7b --dtype Q4_0 --steps 256 --warmup 16 --cpu-seconds 4 --parity-tokens 24 --abi-tokens 8 --no-other-configs > $O/bench_parity.json 2> $O/bench_parity.err; tail -1 $O/bench_parity.json | line parity
  for rep in 1 2; do
    bench_one main_$rep rwkv6-7b Q4_0 --parity-tokens 0
    RWKV_MI_RING_NO_EMBED=1 bench_one noembed_$rep rwkv6-7b Q4_0 --parity-tokens 0
    for v in ${VARIANTS:-f e}; do RWKV_LIB_DIR=lib_$v bench_one ${v}_$rep rwkv6-7b Q4_0 --parity-tokens 0; done
  done
  ;;
seq)    # sequence mode: default (exact) and opt-in (fast) arms -- tests, then the prefill lines with the whole-model gate
  timeout 1500 python -X faulthandler -m pytest tests/test_gpu_prefill.py tests/test_gpu_prefill_fast.py tests/test_gpu_seq_f16.py tests/test_gpu_real_geometry.py tests/test_gpu_reference_programs.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
  pline() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); p=d.get('parity',{}); f=d.get('fast_arms',{})
print('\$1', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms/pass', 'gemm TOP/s', round(r.get('achieved',0),1), '| fast arms', round(f.get('tokens_per_s',0),1), 'tok/s', round(f.get('gemm_TOPs_in_launches',0),1), 'TOP/s | parity', p.get('equal'), p.get('default_arms_bit_identical'), json.dumps(p.get('fast_arms',{}))[:700], flush=True)
" | sed "s/^\\\$1/$1/"; }
  timeout 500 python bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 25 --parity-tokens ${PARITY:-128} > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err; tail -1 $O/prefill_1b6_q4_0.json | pline 1b6; tail -2 $O/prefill_1b6.err
  timeout 500 python bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err; tail -1 $O/prefill_7v_2b9_q5_1.json | pline 2b9; tail -2 $O/prefill_2b9.err
  ;;
sweep)  # the ring kernel's run-time knobs once more (records taken ahead in G and the deferred quantisation moved the balance): one box, alternating
  export RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring
  for rep in 1 2; do
    bench_one base_$rep rwkv6-7b Q4_0 --parity-tokens 0 --no-profile
    for kv in NAP=0 NAP=1 NAP=4 LOOK=2 LOOK=3 THIN=8 THIN=24 INFLIGHT=32 INFLIGHT=40 HTHIN=16 HTHIN=24 BURST=12 BURST=16 HEAD_WG=0 HEAD_WG=64 HEAD_WG=192; do
      env RWKV_MI_RING_$kv bash -c "$(declare -f line bench_one); O=$O; STEPS=${STEPS:-256}; bench_one ${kv}_$rep rwkv6-7b Q4_0 --parity-tokens 0 --no-profile"
    done
  done
  ;;
suite)
  ( timeout 2700 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15 ) > $O/pytest.txt; cat $O/pytest.txt
  grep -q " passed" $O/pytest.txt || { echo "SUITE DID NOT FINISH: no evidence recorded"; exit 1; }
  ;;
esac
