#!/bin/bash
# VGPR / SGPR / scratch of the k_fuse instances (device-only compile of f3d_fuse.hip, ~20 s)
cd "$(dirname "$0")/../3d-point-cloud-segmentation-using-2d-img-segmentation_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. "$@" \
  -S --cuda-device-only -Rpass-analysis=kernel-resource-usage f3d_fuse.hip -o /tmp/f3d_fuse.s 2>&1 |
  python3 -c "
import sys,re
name=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: name=m.group(1); vals={}; continue
    m=re.search(r'(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)',line)
    if m and name:
        vals[m.group(1).split()[0]]=m.group(2)
        if m.group(1).startswith('Occupancy') and ('k_fuseI' in name or 'audit' in name):
            short=re.sub(r'_ZN12_GLOBAL__N_1\d+','',name)[:40]
            print(short, vals)
"
