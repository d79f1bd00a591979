#!/bin/bash
# VGPR / spill / scratch / occupancy of every instantiation in a .hip file: tools/kernel_resources.sh kernels2.hip [extra flags]
cd "$(dirname "$0")/../cannoles.jl_amd/csrc" || exit 1
f=${1:-kernels2.hip}; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c -o /tmp/kres.o "$f" 2>&1 | python3 -c "
import sys,re
for line in sys.stdin:
    if 'error' in line: print(line)
    m=re.search(r'Function Name: (\S+)',line)
    if m: print(); print(m.group(1)[:74],end=' ')
    for k,rx in (('VGPR',r' VGPRs: (\d+)'),('AGPR',r' AGPRs: (\d+)'),('vspill',r'VGPRs Spill: (\d+)'),('sspill',r'SGPRs Spill: (\d+)'),('scratch',r'ScratchSize[^:]*: (\d+)'),('occ',r'Occupancy[^:]*: (\d+)'),('lds',r'LDS Size[^:]*: (\d+)')):
        m=re.search(rx,line)
        if m: print(k+'='+m.group(1),end=' ')
print()
"
