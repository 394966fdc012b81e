cd /root/repo; O=$PWD/gpurun_out/r06x3; mkdir -p $O; R=$PWD
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
python bench.py --config rwkv7-2b9 --dtype Q5_1 --steps 2 --warmup 1 --cpu-seconds 0 --abi-tokens 0 --no-profile --no-other-configs > /dev/null 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --config rwkv7-2b9 --dtype Q5_1 --mode prefill --steps 3 --warmup 1 --cpu-seconds 0 --parity-tokens 0 > /dev/null 2> $O/prof.err
python - $O/prof/p_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:86]:86s} calls {r['Calls']:>6s} tot_ms {float(r['TotalDurationNs'])/1e6:8.3f} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
