#!/bin/bash
# VGPRs / scratch / occupancy of the conv_kernel instantiations under extra flags (CPU only):  scripts/conv_regs.sh [-D...] [filter-regex]
cd /root/repo/phendiff_amd/csrc
FILTER='unsigned short, 3, 1, 8, 32'
ARGS=()
for a in "$@"; do case "$a" in -*) ARGS+=("$a");; *) FILTER="$a";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -c -Rpass-analysis=kernel-resource-usage "${ARGS[@]}" conv_igemm.hip -o /dev/null 2>&1 \
 | python3 -c "
import re,sys,subprocess
rows={};cur=None
for l in sys.stdin:
    m=re.search(r'remark: Function Name: (\S+)',l)
    if m: cur=rows.setdefault(m.group(1),{}); continue
    if cur is None: continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('sgpr',r'TotalSGPRs: (\d+)')):
        m=re.search(pat,l)
        if m and k not in cur: cur[k]=int(m.group(1))
names=sorted(rows)
dem=subprocess.run(['c++filt'],input='\n'.join(names),stdout=subprocess.PIPE,text=True).stdout.splitlines()
for n,d in zip(names,dem):
    if re.search(sys.argv[1],d): r=rows[n]; print(f\"{d[:110]:110s} v{r.get('vgpr')} s{r.get('sgpr')} scratch {r.get('scratch')} occ {r.get('occ')}\")
" "$FILTER"
