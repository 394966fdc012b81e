#!/bin/bash
# Second evidence run of round 4 (after the sequence-mode changes: k_wkv7_seq's broadcast chain, k_mmf16_combine; the ring kernel and its
# quotes are those of tools/gpu_final_r4.sh / profiles/r04y_*): the GPU suite, the prefill lines, the matrix-pipe PMC passes re-stamped.
set -u
cd "$(dirname "$0")/.."
T=${1:-r04v}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
( timeout 420 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error" | tail -8 ) > $O/pytest.txt; cat $O/pytest.txt
B="timeout 150 python bench.py"
$B --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 5 --warmup 2 --cpu-seconds 12 --parity-tokens 1024 > $O/prefill_1b6_q4_0.json 2> $O/prefill_1b6.err
cd /tmp
timeout 100 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/$O/pmc_mfma -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma.err
timeout 100 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA --output-format csv -d $R/$O/pmc_mfma2 -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/pmc_mfma2.err
cd $R
PSTAMP=$(python -c "import sys; sys.argv=['x']; import importlib.util as u; s=u.spec_from_file_location('b','bench.py'); b=u.module_from_spec(s); s.loader.exec_module(b); print(b.prefill_source_stamp())")
cp profiles/pmc_mfma.json $O/pmc_mfma.json
python tools/pmc_mfma_summary.py $O rwkv6-1b6:Q4_0:prefill $PSTAMP $O/pmc_mfma.json > $O/pmc_mfma_summary.txt 2>&1; cat $O/pmc_mfma_summary.txt
rm -f /tmp/synthetic-rwkv6-1b6*
$B --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 6 --parity-tokens 128 > $O/prefill_7v_2b9_q5_1.json 2> $O/prefill_2b9.err
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_v7_prefill -o prefill -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $R/$O/rocprof_v7_prefill.err
cd $R
for f in $O/prefill_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    print(sys.argv[1], round(d["value"],1), d["unit"], "ms/step", round(d["ms_per_step"],3), "parity", (d.get("parity") or {}).get("equal"), "mfma_busy", bool(r.get("mfma_busy")))
except Exception as e:
    print("unreadable", sys.argv[1], e)
PY
done
