#!/bin/bash
# compile ring_v6.hip to asm (device only) and print the register / scratch budget of every k6_ring instantiation; extra flags: "$@"
cd "$(dirname "$0")/../rwkv.cpp_amd"
OUT=${RING_S:-/tmp/ring1.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRWKV_SHARED -DRWKV_BUILD -I../include -Icsrc -S --cuda-device-only csrc/ring_v6.hip -o $OUT "$@" 2>&1 | grep -v hip-link | grep -E "error|warning: v" | head -20
python3 - $OUT <<'PY'
import re,sys
t=open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(_Z\S*k6_ring\S*)\n(.*?)\.wavefront_size', t, re.S):
    nm=m.group(1); body=m.group(2)
    g=lambda k: (re.search(r'\.'+k+r':\s+(\d+)', body) or [0,'?'])[1]
    args=re.search(r'k6_ringILi(\d+)ELi(\d+)', nm)
    print(f"fmt {args.group(1)} ept {args.group(2)}: vgpr {g('vgpr_count')} sgpr {g('sgpr_count')} spill_v {g('vgpr_spill_count')} spill_s {g('sgpr_spill_count')} scratch {g('private_segment_fixed_size')}")
PY
