#!/bin/bash
# matrix-pipe counters of k_mmfx_seq (the exact F16 / F32 sequence GEMM on v_mfma_f32_16x16x4_f32) over a 1024-token pass of an FP32 and an FP16 1.6B file
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r06fx}; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
R=$PWD
for dt in FP32 FP16; do
  timeout 300 python bench.py --config rwkv6-1b6 --dtype $dt --mode prefill --steps 2 --warmup 1 --cpu-seconds 0 > $O/prefill_$dt.json 2> $O/prefill_$dt.err
  mkdir -p $O/$dt
  ( cd /tmp
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/$O/$dt/pmc_mfma -o p -- python $R/bench.py --config rwkv6-1b6 --dtype $dt --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/$dt/pmc_mfma.err
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --output-format csv -d $R/$O/$dt/pmc_mfma2 -o p -- python $R/bench.py --config rwkv6-1b6 --dtype $dt --mode prefill --steps 1 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/$dt/pmc_mfma2.err
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$dt/prof -o p -- python $R/bench.py --config rwkv6-1b6 --dtype $dt --mode prefill --steps 2 --warmup 1 --cpu-seconds 0 > /dev/null 2> $R/$O/$dt/prof.err )
  echo "== $dt"; python tools/pmc_mfma_summary.py $O/$dt rwkv6-1b6:$dt:prefill none $O/pmc_mfma_fx.json k_mmfx_seq
  python - $O/$dt/prof/p_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>6s} tot_ms {float(r['TotalDurationNs'])/1e6:8.3f} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
done 2>&1 | tee $O/pmc_mfma_fx_summary.txt
