#!/usr/bin/env python3
"""Kernel resource usage from hipcc -Rpass-analysis=kernel-resource-usage remarks (make -C pytracer_amd/csrc resource-usage 2>&1 |
python3 tools/resource_usage.py /dev/stdin [regex]): demangled name, VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, static LDS."""
import re,sys,subprocess
txt=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else 'k_shade'
blocks=re.split(r"remark: Function Name: ", txt)[1:]
for b in blocks:
    name=b.split()[0]
    dem=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    if not re.search(pat, dem): continue
    def g(k):
        m=re.search(k+r": (\d+)", b); return m.group(1) if m else '?'
    print(dem.split('(')[0][:70], 'VGPR',g('VGPRs'),'AGPR',g('AGPRs'),'scratch',g('ScratchSize \[bytes/lane\]'),'occ',g('Occupancy \[waves/SIMD\]'),'LDS',g('LDS Size \[bytes/block\]'))
