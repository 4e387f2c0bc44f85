#!/usr/bin/env python3
"""Dev tool: outline of one kernel in a hipcc -save-temps .s file: labels, branches, barriers, waits, scratch traffic,
LDS-DMA, with the running count of MFMAs / ds_reads - enough to see what the scheduler did to a hand-placed loop.
usage: isa_outline.py file.s <substring of the mangled kernel name> [--full]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
mf = ds = vm = 0
for i in range(start, end + 1):
    t = lines[i].strip()
    if t.startswith('v_mfma'): mf += 1
    if t.startswith('ds_read'): ds += 1
    if re.match(r'(scratch_|s_barrier|s_waitcnt|s_cbranch|s_branch|buffer_load|buffer_store|global_load|global_store|s_setprio|\.LBB)', t):
        if t.startswith('s_waitcnt') and 'vmcnt' not in t and '--full' not in sys.argv: continue
        if t.startswith('s_setprio') and '--full' not in sys.argv: continue
        print(f"{i - start:6d} mfma={mf:4d} ds={ds:4d}  {t[:100]}")
