#!/bin/bash
# Copies the summaries of an evidence run (tools/gpu_final.sh <tag>) from gpurun_out/<tag>/ into profiles/ under the round's naming.
set -u
cd "$(dirname "$0")/.."
T=${1:?tag}; O=gpurun_out/$T; P=profiles
for f in $O/bench_*.json $O/prefill_*.json $O/chain_*.json $O/pipeline_*.json; do [ -s "$f" ] && cp "$f" $P/${T}_$(basename "$f"); done
stats() { f=$(find $O/$1 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $P/${T}_$2_kernel_stats.csv; }
stats prof_7b decode_7b; stats prof_v7 decode_v7; stats prof_v4 decode_v4; stats prof_prefill prefill_1b6_q4_0; stats prof_v7_prefill prefill_v7_2b9_q5_1; stats prof_1b6 decode_1b6
for c in fetch write; do f=$(find $O/pmc_$c -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && python - "$f" $P/${T}_pmc_$c.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k6_ring" in r["Kernel_Name"] or "k6_mega" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id"])
w.writeheader()
for r in rows: w.writerow({k: (r[k][:60] if k == "Kernel_Name" else r[k]) for k in w.fieldnames})
PY
done
for f in ring_phase_trace_7b.txt ring_head_trace_7b.txt pytest.txt head.txt hop_ab.txt prefill_fuse_ab.txt; do [ -s $O/$f ] && cp $O/$f $P/${T}_$f; done
ls -la $P | grep "${T}_" | awk '{print $5, $9}'
