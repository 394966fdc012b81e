#!/bin/bash
# kernel resource usage of one .hip file: tools/kres.sh csrc/ring_v6.hip [pattern]
cd "$(dirname "$0")/../rwkv.cpp_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRWKV_SHARED -DRWKV_BUILD -fvisibility=hidden -I../include -Icsrc -Wall -Wno-unused-function $EXTRA \
  -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $1 -o build/$(basename ${1%.*}).o 2>&1 | python3 -c "
import sys,re
pat=sys.argv[1] if len(sys.argv)>1 else ''
cur=None
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}
    for k in ('TotalSGPRs','VGPRs','AGPRs','ScratchSize \[bytes/lane\]','LDS Size'):
        m=re.search(k+r': (\d+)',l)
        if m and cur: d[k.split()[0]]=m.group(1)
    if 'LDS Size' in l and cur and pat in cur: print(cur[:70], d)
" "${2:-}"
