"""Instruction mix of the main (MFMA) loop of one kernel in a `hipcc -S` dump: python tools/loopmix.py file.s <mangled-name-substring>"""
import re, sys
from collections import Counter
L = open(sys.argv[1]).read().split('\n'); key = sys.argv[2]
start = [i for i, l in enumerate(L) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E') and ':' in l][0]
end = [i for i, l in enumerate(L) if '.amdhsa_kernel' in l and key in l][0]
body = L[start:end]
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
# loop = from the last "Loop Header" label before the first barrier-bearing mfma region to the last backward branch
hdr = [i for i, l in enumerate(body) if 'Loop Header' in l and 'Depth=1' in l]
h = max(i for i in hdr if i < mf[-1])
lab = body[h].split(':')[0]
tail = max(i for i, l in enumerate(body) if re.search(r's_c?branch\w* ' + re.escape(lab) + r'\b', l))
loop = body[min(h, tail):max(h, tail) + 1] if tail > h else body[tail:h]
c = Counter()
for l in loop:
    l = l.strip()
    if not l or l[0] in ';.': continue
    op = l.split()[0]
    c['mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else op] += 1
print(key, dict(c))
