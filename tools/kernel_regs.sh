#!/bin/bash
# tools/kernel_regs.sh [file.hip] [filter]: VGPR/SGPR/spill/occupancy of every kernel of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage)
F=${1:-zoic_amd/csrc/kolb_pool.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude $ZOIC_EXTRA_HIPCC_FLAGS -c $F -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import re,sys
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur=None; rows={}
for l in sys.stdin:
    if 'error' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('TotalSGPRs','VGPRs','Occupancy \[waves/SIMD\]','SGPRs Spill','VGPRs Spill','LDS Size \[bytes/block\]'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: rows[cur][k]=m.group(1)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.split('(')[0].replace('void zoic::','')
    if flt in name: print('%-40s vgpr %3s sgpr %3s occ %s spill s%s v%s' % (name, v.get('VGPRs'), v.get('TotalSGPRs'), v.get('Occupancy \[waves/SIMD\]'), v.get('SGPRs Spill'), v.get('VGPRs Spill')))
" "$2"
