#!/bin/bash
# usage: tools/kernel_resources.sh <unit.hip> [extra hipcc flags]: VGPRs / spills / LDS / occupancy per kernel (gfx950)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /tmp/_kr.o 2>&1 | \
python3 -c "
import sys,re,subprocess
cur=None; d={}
for l in sys.stdin:
    m=re.search(r'remark: Function Name: (\S+)',l)
    if m:
        cur=m.group(1); d={}
        continue
    m=re.search(r'remark:\s+([A-Za-z][^:]*): (\S+)',l)
    if m and cur:
        d[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith('LDS Size'):
            name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip().split('(')[0]
            print('%-72s VGPR %3s AGPR %3s scratch %4s SGPR %3s occ %s LDS %s'%(name[-72:],d.get('VGPRs'),d.get('AGPRs'),d.get('ScratchSize [bytes/lane]'),d.get('TotalSGPRs'),d.get('Occupancy [waves/SIMD]'),d.get('LDS Size [bytes/block]')))
"
