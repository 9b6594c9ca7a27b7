"""Instruction mix of the hottest loop (the one holding the MFMAs) of a kernel in a hipcc -S dump."""
import re, sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n'); key = sys.argv[2]
start = [i for i, l in enumerate(L) if key in l and l.rstrip().endswith(key.join(['', '']) ) or (key in l and '; @' in l)][0]
end = [i for i, l in enumerate(L) if key in l and '.amdhsa_kernel' in l][0]
body = L[start:end]
idx = [i for i, l in enumerate(body) if 'v_mfma' in l]
lab = None
for i in range(idx[-1], len(body)):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', body[i])
    if m:
        pos = [j for j, l in enumerate(body) if l.startswith(m.group(1) + ':')]
        if pos and pos[0] < idx[0]:
            lab = (pos[0], i); break
loop = body[lab[0]:lab[1] + 1]
c = Counter()
for l in loop:
    l = l.strip()
    if not l or l[0] in ';.': continue
    op = l.split()[0]
    c['mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else op] += 1
print(dict(c)); open('/tmp/loop.s', 'w').write('\n'.join(loop))
