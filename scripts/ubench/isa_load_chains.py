#!/usr/bin/env python3
"""Static scan of a device listing (hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o gpx.s gigapaxos_amd/csrc/
gpx_engine.hip) for loads that travel one at a time: per kernel the longest run of (one vector load, s_waitcnt
vmcnt(0)) pairs.  A long run is either pointer chasing (legitimate) or independent loads the compiler serialised -
the listing decides which.  scripts/ubench/isa_sequence.py prints a kernel's memory instructions / waits / barriers
in order.  Usage: isa_load_chains.py gpx.s"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
kern = None
seqs = {}
for l in src:
    m = re.match(r'^(_Z\S*):', l)
    if m:
        kern = m.group(1); seqs[kern] = []; continue
    if kern is None: continue
    if l.startswith('.Lfunc_end'): kern = None; continue
    m = re.search(r'\b(global_load\w*|flat_load\w*|scratch_load\w*|s_waitcnt [^;]*|s_barrier|global_store\w*|flat_store\w*|ds_\w+|global_atomic\w*)', l)
    if m: seqs[kern].append(m.group(1))
import subprocess
rows = []
for k, s in seqs.items():
    # chains: LD (1 load) followed by wait vmcnt(0), repeated
    best = cur = 0; i = 0; nld = 0
    while i < len(s):
        if re.match(r'(global|flat)_load', s[i]):
            nld += 1
            if i + 1 < len(s) and s[i+1].startswith('s_waitcnt') and 'vmcnt(0)' in s[i+1]:
                cur += 1; best = max(best, cur); i += 2; continue
            cur = 0
        elif s[i].startswith('s_waitcnt') and 'vmcnt' not in s[i]:
            pass
        else:
            cur = 0
        i += 1
    rows.append((best, nld, k))
rows.sort(reverse=True)
for b, n, k in rows[:40]:
    name = subprocess.run(['c++filt', '-p', k], stdout=subprocess.PIPE, text=True).stdout.strip()
    print(f"{b:3d} longest chain of (one load, wait)   {n:4d} loads   {name[:80]}")
