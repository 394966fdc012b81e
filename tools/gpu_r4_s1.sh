#!/bin/bash
# round 4, step 1: the whole GPU suite on the refactored host side (ring streams shared per model, hooks library, chain mailboxes, new
# real-geometry slices) + a same-box baseline of the headline line and its phase trace
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp RWKV_BENCH_DIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25 ) > $O/pytest.txt; cat $O/pytest.txt
timeout 500 python bench.py --steps 128 --warmup 16 > $O/bench_7b_q4_0.json 2> $O/bench_7b_q4_0.err; tail -c 1500 $O/bench_7b_q4_0.json
RWKV_MI_RING_LTRACE=/tmp/lt.bin timeout 200 python tools/trace_ring.py rwkv6-7b 5 > $O/ring_phase_trace_7b.txt 2> $O/trace.err; head -40 $O/ring_phase_trace_7b.txt
