#!/bin/bash
# tuning only: VGPR / spill / LDS summary of every kernel in a .hip file (kernel-resource-usage remarks)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/kres.o "${SRC:-/root/repo/laudnet_amd/csrc/ldn_conv_image.hip}" 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l) or re.search(r' Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; continue
    for k in ['VGPRs','AGPRs','ScratchSize \[bytes/lane\]','VGPR Spill','SGPR Spill','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]']:
        m=re.search(r'    '+k+r': (\d+)',l)
        if m and cur is not None:
            cur[k]=m.group(1)
            if k.startswith('LDS'):
                print(cur['name'][:70], 'V',cur.get('VGPRs'),'A',cur.get('AGPRs'),'scr',cur.get('ScratchSize \[bytes/lane\]'),'occ',cur.get('Occupancy \[waves/SIMD\]'))
"
