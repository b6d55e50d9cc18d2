#!/bin/bash
# registers / LDS / occupancy of every kernel of the given sources from the compiler's resource-usage remarks (no GPU needed)
# usage: tools/kernel_resources.sh [file.hip ...]   (default: all of csrc/)
cd "$(dirname "$0")/../segment-anything-in-nerf_amd/csrc"
files=${@:-*.hip}
for f in $files; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I ../../include $EXTRA -c $f -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark: (.*?):\s*(.*?) \[-Rpass', l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=='Function Name': cur=v; rows[cur]={}
    elif cur: rows[cur][k]=v
names=subprocess.run(['c++filt']+list(rows),capture_output=True,text=True).stdout.strip().split('\n') if rows else []
for n,(_,r) in zip(names,rows.items()):
    dn=re.sub(r'\(.*','',n).replace('void snf::','')
    print('%-84s V %4s A %4s occ %2s spill %3s LDS %6s'%(dn[:84], r.get('VGPRs'), r.get('AGPRs'), r.get('Occupancy [waves/SIMD]'), r.get('VGPRs Spill'), r.get('LDS Size [bytes/block]')))
"; done; rm -f /tmp/kr_$$.o
