#!/bin/bash
# Evidence run of a round (the GPU test suite runs separately: `pytest tests -m gpu`): bench lines of every BASELINE configuration,
# rocprofv3 kernel stats, PMC traffic passes, the ring kernel's phase / loader trace. Outputs under gpurun_out/$1; the summaries that
# matter are copied to profiles/ afterwards (tools/collect_profiles.sh $1).
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03z}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
B="timeout 400 python bench.py"
$B --steps 256 --warmup 16 > $O/bench_7b_q4_0.json 2> $O/bench_7b_q4_0.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_7b -o decode -- python $R/bench.py --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_7b.err
if [ -z "${RWKV_FINAL_SKIP_PMC:-}" ]; then
RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/pmc_fetch.err
RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 --no-other-configs > /dev/null 2> $R/$O/pmc_write.err
fi
cd $R
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err
timeout 200 python tools/trace_head.py rwkv6-7b > $O/ring_head_trace_7b.txt 2>> $O/trace.err
rm -f /tmp/synthetic-rwkv6-7b-Q4_0*
$B --config rwkv6-1b6 --dtype Q4_0 --steps 256 --cpu-seconds 5 > $O/bench_1b6_q4_0.json 2> $O/bench_1b6.err
$B --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 25 --parity-tokens 1024 > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err
timeout 300 python bench.py --gpus 2 --chain --chain-devices 0,0 --config rwkv6-1b6 --steps 128 --warmup 8 > $O/chain_2stages_1b6.json 2> $O/chain.err
RWKV_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config rwkv6-1b6 --steps 32 --warmup 4 > $O/pipeline_gloo_2ranks_1gpu.json 2> $O/pipeline_gloo.err
$B --config rwkv7-2b9 --dtype Q5_1 --steps 128 --cpu-seconds 5 > $O/bench_7v_2b9_q5_1.json 2> $O/bench_2b9.err
$B --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err
$B --config rwkv4-169m --dtype Q5_1 --steps 256 --cpu-seconds 5 > $O/bench_4_169m_q5_1.json 2> $O/bench_169m.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7 -o decode -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --steps 48 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_v7.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7_prefill -o prefill -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_v7_prefill.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v4 -o decode -- python $R/bench.py --config rwkv4-169m --dtype Q5_1 --steps 48 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_v4.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_prefill -o prefill -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_prefill.err
cd $R
rm -f /tmp/synthetic-rwkv6-1b6* /tmp/synthetic-rwkv7* /tmp/synthetic-rwkv4*
$B --config rwkv6-7b --dtype Q8_0 --steps 64 --cpu-seconds 5 > $O/bench_7b_q8_0.json 2> $O/bench_7b_q8.err
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(d["metric"], round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "path", d["config"].get("decode_path"), d["config"].get("persist_kind"), "roof", round(r.get("frac",0),4), "avg_us", round(r.get("avg_launch_us",0),1), "parity", d.get("parity",{}).get("equal"), "abi", round(d.get("abi",{}).get("tokens_per_s",0),1), "cpu", round(d.get("cpu_baseline",{}).get("value",0),2), "load", d.get("load",{}).get("cold_seconds"), d.get("load",{}).get("warm_seconds"), "multi", d.get("multi_stream",{}).get("tokens_per_s_aggregate"))
except Exception as e:
    print("unreadable", e)
PY
done
[ "${RWKV_FINAL_SKIP_SUITE:-0}" = 1 ] || { ( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5 ) > $O/pytest.txt; cat $O/pytest.txt; }
