#!/bin/bash
# per-kernel times of the 1.6B Q4_0 1024-token pass (rocprofv3 --kernel-trace --stats), default build switches vs "$2" env assignments
cd "$(dirname "$0")/.."; T=${1:-r06pp}; O=$PWD/gpurun_out/$T; mkdir -p $O; R=$PWD
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
python bench.py --config rwkv6-1b6 --steps 2 --warmup 1 --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs > /dev/null 2>&1
cd /tmp
for v in new old; do
  if [ $v = old ]; then export RWKV_MI_NO_EPI_QUANT=1 RWKV_MI_NO_GN_QUANT=1 RWKV_MI_NO_MIX_QUANT=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --config rwkv6-1b6 --dtype Q4_0 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $O/prof_$v.err
  python - $O/prof_$v/p_kernel_stats.csv <<'PY' | tee $O/stats_$v.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} tot_ms {float(r['TotalDurationNs'])/1e6:8.3f} avg_us {float(r['AverageNs'])/1e3:8.2f}")
PY
done
