#!/usr/bin/env python3
"""Loop census of a kernel's ISA: python profiles/isa_loops.py file.s <substring of the kernel's mangled name>.
For every backward branch: line span and counts of MFMA / scratch / LDS / VALU / waitcnt / barrier instructions in it."""
import re
import sys

s = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
i = 0
while i < len(s):
    m = re.match(r'^(\S*%s\S*):' % re.escape(pat), s[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i
    while j < len(s) and '.end_amdhsa_kernel' not in s[j] and not s[j].startswith('.Lfunc_end'):
        j += 1
    body = s[i:j]
    print(name, len(body), 'lines; scratch', sum('scratch_' in x for x in body), 'mfma', sum('v_mfma' in x for x in body))
    labels = {}
    for k, l in enumerate(body):
        lm = re.match(r'^\s*(\.LBB\d+_\d+):', l)
        if lm:
            labels[lm.group(1)] = k
    for k, l in enumerate(body):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            seg = body[labels[mm.group(1)]:k]
            print('  loop %s lines %d-%d: mfma %d scratch %d ds_read %d ds_write %d valu %d waitcnt %d barrier %d buffer/global %d' % (
                mm.group(1), labels[mm.group(1)], k, sum('v_mfma' in x for x in seg), sum('scratch_' in x for x in seg),
                sum('ds_read' in x for x in seg), sum('ds_write' in x for x in seg),
                sum(bool(re.match(r'\s*v_(?!mfma)', x)) for x in seg), sum('s_waitcnt' in x for x in seg),
                sum('s_barrier' in x for x in seg), sum(bool(re.match(r'\s*(buffer_|global_)', x)) for x in seg)))
    i = j
