#!/bin/bash
# GPU session 1 of round 2: parity tests, bench lines of every BASELINE config on the current build, nt A/B, prefill "before" number.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -25 ) > $O/pytest.txt
echo "pytest done: $(tail -1 $O/pytest.txt)"
B="timeout 600 python bench.py"
$B --steps 128 --warmup 16 > $O/bench_7b_q4_0_nt.json 2> $O/bench_7b_q4_0_nt.err; tail -c 600 $O/bench_7b_q4_0_nt.json
RWKV_LIB_DIR=lib_plain $B --steps 128 --warmup 16 --cpu-seconds 0 --abi-tokens 0 > $O/bench_7b_q4_0_plain.json 2> $O/bench_7b_q4_0_plain.err
RWKV_MI_NO_MEGA=1 $B --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 > $O/bench_7b_q4_0_fused_nt.json 2>/dev/null
RWKV_MI_NO_MEGA=1 RWKV_LIB_DIR=lib_plain $B --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 > $O/bench_7b_q4_0_fused_plain.json 2>/dev/null
rm -f /tmp/synthetic-rwkv6-7b-Q4_0*
$B --config rwkv6-1b6 --dtype Q4_0 --steps 256 --cpu-seconds 6 > $O/bench_1b6_q4_0.json 2> $O/bench_1b6.err
$B --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 6 > $O/prefill_1b6_q4_0_before.json 2> $O/prefill_1b6.err
$B --config rwkv7-2b9 --dtype Q5_1 --steps 128 --cpu-seconds 6 > $O/bench_7v_2b9_q5_1.json 2> $O/bench_2b9.err
$B --config rwkv4-169m --dtype Q5_1 --steps 256 --cpu-seconds 6 > $O/bench_4_169m_q5_1.json 2> $O/bench_169m.err
rm -f /tmp/synthetic-rwkv6-1b6* /tmp/synthetic-rwkv7* /tmp/synthetic-rwkv4*
$B --config rwkv6-7b --dtype Q8_0 --steps 64 --cpu-seconds 6 > $O/bench_7b_q8_0.json 2> $O/bench_7b_q8.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(d["metric"], round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "path", d["config"].get("decode_path"), "roof", round(r.get("frac",0),4), "avg_us", round(r.get("avg_launch_us",0),1), "parity", d.get("parity",{}).get("equal"), "abi", round(d.get("abi",{}).get("tokens_per_s",0),1), "cpu", round(d.get("cpu_baseline",{}).get("value",0),2))
except Exception as e:
    print("unreadable", e)
PY
done
