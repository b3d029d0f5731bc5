#!/bin/bash
# tools/kres.sh <file.hip> <kernel substring>: registers / spills / LDS / occupancy of the kernels of one translation unit
# (device-only compile with -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../supervised-depth-estimation-from-polarized-images_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --cuda-device-only -c "$1" -o /tmp/kres_dev.o ${PD_DEFS} \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r'remark: (?:\S+ )?Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark: +(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',ln)
    if m and cur is not None: cur[m.group(1)]=m.group(2)
for r in rows:
    d=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    if '$2' in d:
        d=re.sub(r'\(anonymous namespace\)::','',d); d=re.sub(r'\(.*','',d)
        print(f\"{d[:70]:70s} vgpr {r.get('VGPRs')} agpr {r.get('AGPRs')} vspill {r.get('VGPRs Spill')} sspill {r.get('SGPRs Spill')} scratch {r.get('ScratchSize [bytes/lane]')} lds {r.get('LDS Size [bytes/block]')} occ {r.get('Occupancy [waves/SIMD]')}\")
"
