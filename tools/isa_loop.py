#!/usr/bin/env python3
"""Prints the instruction histogram / listing of the loop that contains a marker instruction in one
kernel of /tmp/qpn_isa/<file>.s (developer aid for counting issue slots)."""
import sys, collections
f, kern, marker = sys.argv[1], sys.argv[2], sys.argv[3]
L = open(f).read().split('\n')
st = [i for i, x in enumerate(L) if x.startswith(kern)][0]
en = [i for i, x in enumerate(L) if x.startswith('.Lfunc_end') and i > st][0]
K = L[st:en]
mk = [i for i, x in enumerate(K) if marker in x]
hd = [i for i, x in enumerate(K) if 'Loop Header' in x]
m = mk[-1] if len(sys.argv) < 5 else mk[int(sys.argv[4])]
a = max(h for h in hd if h <= m)
label = K[a].split(':')[0]
import re
# loop latch: last branch back to the header, following intermediate flow labels that fall into it
tgt = {label}
for i in range(a - 1, max(a - 40, 0), -1):
    if K[i].startswith('.LBB'): tgt.add(K[i].split(':')[0])
    elif K[i].strip().startswith('s_branch') or K[i].strip().startswith('s_endpgm'): break
b = max(i for i, x in enumerate(K) if any(x.strip().endswith(' ' + t) for t in tgt) and 'branch' in x) + 1
body = [x for x in K[a:b] if x.strip() and not x.strip().startswith(';')]
c = collections.Counter(x.split()[0] for x in body if not x.startswith('.'))
print(len(body), 'lines in loop'); print(sorted(c.items(), key=lambda t: -t[1])[:40])
if '-l' in sys.argv: print('\n'.join(body))
