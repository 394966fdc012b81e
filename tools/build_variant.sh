#!/bin/bash
# A second librwkv.so with ONE object built with extra flags (A/B runs on one box: tools/gpu_r5.sh ab):
#   tools/build_variant.sh lib_b persist_v47 -DP47_PARK=0
#   tools/build_variant.sh lib_b ring_v6 -DR6_WATCH_SPREAD=0
set -eu
cd "$(dirname "$0")/../rwkv.cpp_amd"
D=$1; SRC=$2; shift 2
mkdir -p build_$D $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 "$@" -O3 -std=c++17 -ffp-contract=off -fPIC -DRWKV_SHARED -DRWKV_BUILD -fvisibility=hidden -I../include -Icsrc -Wall -Wno-unused-function -Wno-unused-variable -c csrc/$SRC.hip -o build_$D/$SRC.o
OBJS=$(for o in pipeline runner format quantize api kernels model engine fused_v6 mega_v6 ring_v6 persist_v47 prefill prefill_fast fused_v7 sampling; do if [ $o = $SRC ]; then echo build_$D/$o.o; else echo build/$o.o; fi; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=csrc/rwkv.map -o $D/librwkv.so $OBJS -ldl
ls -la $D/librwkv.so
