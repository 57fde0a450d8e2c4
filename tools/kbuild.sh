#!/bin/bash
# kbuild.sh <kernel file stem> [extra hipcc flags]: compile one csrc/*.hip for gfx950 with -save-temps and print, per kernel,
# LDS bytes / scratch bytes / VGPRs (offline check before a GPU run)
set -e
cd /root/repo/yolo2_light_amd/csrc
stem=$1; shift
mkdir -p /tmp/kbuild
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $stem.hip -o /tmp/kbuild/$stem.o -save-temps=obj
python3 - "$stem" <<'PY'
import re,sys
s=open('/tmp/kbuild/%s-hip-amdgcn-amd-amdhsa-gfx950.s'%sys.argv[1]).read()
for m in re.finditer(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', s, re.S):
    name=re.sub(r'_ZN2yl12_GLOBAL__N_1\d+','',m.group(2)); name=re.sub(r'EEvNS0_\d+\w+$','',name)
    print("%-60s lds %6s scratch %3s sgpr %3s vgpr %3s"%(name[:60],m.group(1),m.group(3),m.group(4),m.group(5)))
PY
