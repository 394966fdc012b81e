#!/bin/bash
# Evidence run of round 4: bench lines of every BASELINE configuration, rocprofv3 kernel stats, PMC traffic passes (+ summary with the
# kernel source stamp), the ring kernel's phase / loader trace, the matrix-pipe PMC passes of the sequence GEMM, the GPU suite.
# Outputs under gpurun_out/$1; tools/collect_profiles.sh $1 copies the summaries into profiles/.
set -u
cd "$(dirname "$0")/.."
T=${1:-r04z}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
git rev-parse HEAD > $O/head.txt 2>/dev/null || true
B="timeout 400 python bench.py"
$B --steps 256 --warmup 16 > $O/bench_7b_q4_0.json 2> $O/bench_7b_q4_0.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_7b -o decode -- python $R/bench.py --steps 64 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_7b.err
RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_fetch -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 > /dev/null 2> $R/$O/pmc_fetch.err
RWKV_MI_NO_AUTOTUNE=1 RWKV_MI_PERSIST=ring RWKV_BENCH_NO_COLD=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pmc_write -o p -- python $R/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --abi-tokens 0 --no-profile --parity-tokens 0 > /dev/null 2> $R/$O/pmc_write.err
cd $R
STAMP=$(python -c "import sys; sys.argv=['x']; import importlib.util as u; s=u.spec_from_file_location('b','bench.py'); b=u.module_from_spec(s); s.loader.exec_module(b); print(b.kernel_source_stamp(2))")
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write k6_ring rwkv6-7b:Q4_0:path2:kind2 $O/pmc_traffic.json $STAMP > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err
timeout 200 python tools/trace_head.py rwkv6-7b > $O/ring_head_trace_7b.txt 2>> $O/trace.err
# the default line once more with the fresh PMC quote in place (what the driver's run will print)
cp $O/pmc_traffic.json profiles/pmc_traffic.json
$B > $O/bench_default.json 2> $O/bench_default.err
rm -f /tmp/synthetic-rwkv6-7b-Q4_0*
$B --config rwkv6-1b6 --dtype Q4_0 --steps 256 --cpu-seconds 5 > $O/bench_1b6_q4_0.json 2> $O/bench_1b6.err
$B --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 25 --parity-tokens 1024 > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/$O/pmc_mfma -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA --output-format csv -d $R/$O/pmc_mfma2 -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_prefill -o prefill -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_prefill.err
cd $R
PSTAMP=$(python -c "import sys; sys.argv=['x']; import importlib.util as u; s=u.spec_from_file_location('b','bench.py'); b=u.module_from_spec(s); s.loader.exec_module(b); print(b.prefill_source_stamp())")
cp profiles/pmc_mfma.json $O/pmc_mfma.json
python tools/pmc_mfma_summary.py $O rwkv6-1b6:Q4_0:prefill $PSTAMP $O/pmc_mfma.json > $O/pmc_mfma_summary.txt 2>&1; cat $O/pmc_mfma_summary.txt
timeout 300 python bench.py --gpus 2 --chain --chain-devices 0,0 --config rwkv6-1b6 --steps 128 --warmup 8 --cpu-seconds 4 > $O/chain_2stages_1b6.json 2> $O/chain.err
$B --config rwkv6-1b6 --dtype FP16 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > $O/prefill_1b6_fp16.json 2> $O/prefill_1b6_fp16.err
rm -f /tmp/synthetic-rwkv6-1b6*
$B --config rwkv7-2b9 --dtype Q5_1 --steps 128 --cpu-seconds 5 > $O/bench_7v_2b9_q5_1.json 2> $O/bench_2b9.err
$B --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 8 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7 -o decode -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --steps 48 --warmup 8 --cpu-seconds 0 --abi-tokens 0 --no-profile > /dev/null 2> $R/$O/rocprof_v7.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7_prefill -o prefill -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_v7_prefill.err
cd $R
rm -f /tmp/synthetic-rwkv7*
$B --config rwkv4-169m --dtype Q5_1 --steps 256 --cpu-seconds 5 > $O/bench_4_169m_q5_1.json 2> $O/bench_169m.err
rm -f /tmp/synthetic-rwkv4*
$B --config rwkv6-7b --dtype Q8_0 --steps 64 --cpu-seconds 5 > $O/bench_7b_q8_0.json 2> $O/bench_7b_q8.err
rm -f /tmp/synthetic-rwkv6-7b*
for f in $O/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(d["metric"], round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "path", d["config"].get("decode_path"), d["config"].get("persist_kind"), "roof", round(r.get("frac",0),4), "avg_us", round(r.get("avg_launch_us",0),1), "traffic", r.get("traffic"), "parity", d.get("parity",{}).get("equal"), "abi", round(d.get("abi",{}).get("tokens_per_s",0),1), "cpu", round(d.get("cpu_baseline",{}).get("value",0),2), "load", d.get("load",{}).get("cold_seconds"), d.get("load",{}).get("warm_seconds"), "multi", d.get("multi_stream",{}).get("tokens_per_s_aggregate"))
except Exception as e:
    print("unreadable", e)
PY
done
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5 ) > $O/pytest.txt; cat $O/pytest.txt
