#!/usr/bin/env python3
"""Are two device listings the same code?  Compares, kernel by kernel, the instruction streams of two
hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only listings (labels, comments and directives normalised).
Used to show that a host-side or template-plumbing change left the GPU-validated kernels untouched when no GPU
is at hand.  Usage: isa_compare.py old.s new.s"""
import re, sys, hashlib
def kernels(path):
    out, cur, name = {}, None, None
    for l in open(path):
        m = re.match(r'^(_Z\S*):', l)
        if m:
            name = m.group(1); cur = []; out[name] = cur; continue
        if cur is None: continue
        if l.startswith('.Lfunc_end'):
            cur = None; continue
        l = l.split(';')[0].rstrip()
        if not l.strip() or l.strip().startswith('.'): 
            if not re.match(r'^\.LBB', l): continue
        l = re.sub(r'LBB\d+_', 'LBB_', l)
        l = re.sub(r'_Z14k_wire_decode1ILi(\d+)E(Lb0E)?', r'_Z14k_wire_decode1ILi\1E', l)
        cur.append(l)
    return out
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
norm = lambda n: re.sub(r'_Z14k_wire_decode1ILi(\d+)E(Lb0E)?', r'_Z14k_wire_decode1ILi\1E', n)
a = {norm(k): v for k, v in a.items()}; b = {norm(k): v for k, v in b.items()}
same = diff = 0
for k in sorted(set(a) | set(b)):
    if k not in a: print("only new:", k[:70]); continue
    if k not in b: print("only old:", k[:70]); continue
    if a[k] == b[k]: same += 1
    else: diff += 1; print("DIFFERS:", k[:90], len(a[k]), len(b[k]))
print(same, "kernels identical,", diff, "differ")
