#!/bin/bash
# compile macenko.hip to asm and report scratch ops inside the fused kernel's sweep loops (blocks with >= 12 v_perm)
cd "$(dirname "$0")/../stainlib_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -S --cuda-device-only macenko.hip -o /tmp/mx.s 2>/dev/null
python3 - <<'PY'
import re,collections
s=open('/tmp/mx.s').read()
for name in ('_ZN2slL7k_fusedILi0ELb1ELb1ELi512EEEvNS_9FusedArgsE','_ZN2slL7k_fusedILi0ELb0ELb1ELi512EEEvNS_9FusedArgsE'):
    i=s.index(name+':'); j=s.index('.Lfunc_end',i)
    body=s[i:j].split('\n')
    blocks=[]; cur=None
    for l in body:
        m=re.match(r'^(\.LBB\d+_\d+):(.*)',l)
        if m: cur=[m.group(1),m.group(2),[]]; blocks.append(cur)
        elif cur is not None and l.strip() and not l.strip().startswith(';'): cur[2].append(l.strip())
    tot=0; lst=[]
    for lab,cm,ins in blocks:
        c=collections.Counter(x.split()[0] for x in ins)
        sc=sum(v for k,v in c.items() if 'scratch' in k)
        if c.get('v_perm_b32',0)>=12: tot+=sc; lst.append(sc)
    print(name[14:40], 'scratch ops in sweep-loop blocks:', tot, lst)
PY
