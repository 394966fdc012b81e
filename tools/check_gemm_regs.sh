#!/bin/bash
# compile prefill.hip to asm and print register stats of the GEMM instantiations
cd /root/repo/rwkv.cpp_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRWKV_SHARED -DRWKV_BUILD -I../include -Icsrc -S --cuda-device-only csrc/prefill.hip -o /tmp/pf1.s $@ 2>&1 | grep -v hip-link | head -20
for f in 2 3 7 8 9; do echo -n "fmt $f: "; grep -A30 "\.name:.*k_mmq_mfmaILi${f}E" /tmp/pf1.s | grep -E "\.vgpr_count|vgpr_spill|private_segment_fixed|agpr" | tr -s ' ' | tr '\n' ' '; echo; done
