#!/usr/bin/env python3
"""Memory instructions, waits, barriers and branches of one kernel of a device listing, in order.
Usage: isa_sequence.py gpx.s <substring of the mangled kernel name>"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
on = False
out = []
for l in src:
    if re.match(r'^_Z\S*:', l):
        on = want in l
        if on: out.append(l.split(':')[0][:70])
        continue
    if not on: continue
    if l.startswith('.Lfunc_end'): on = False
    m = re.search(r'\b(global_load\w*|flat_load\w*|buffer_load\w*|scratch_load\w*|global_store\w*|flat_store\w*|global_atomic\w*|flat_atomic\w*|ds_\w+|s_waitcnt [^;]*|s_barrier|s_cbranch\w*|s_load\w*)', l)
    if m:
        t = m.group(1)
        t = re.sub(r'global_load_|flat_load_', 'LD.', t); t = re.sub(r'global_store_|flat_store_', 'ST.', t)
        t = t.replace('s_waitcnt ', 'W:').replace('s_cbranch_', 'br.')
        out.append(t)
# compress
line = []
for t in out:
    line.append(t)
print(' '.join(line))
