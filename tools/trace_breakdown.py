#!/usr/bin/env python3
"""Dev tool: group a rocprofv3 --kernel-trace csv by (kernel, grid) and print calls / mean / total duration.
usage: trace_breakdown.py <dir> [skip_first_fraction]"""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * skip):]
acc = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\s+", "", r["Kernel_Name"].replace("void ", ""))[:70]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", ""))
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = acc.setdefault(key, [0, 0])
    a[0] += 1; a[1] += dur
tot = sum(a[1] for a in acc.values())
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:70s} grid={k[1]:>9s} wg={k[2]:>5s} calls={a[0]:6d} mean={a[1]/a[0]/1e3:9.1f} us total={a[1]/1e6:9.2f} ms {100*a[1]/tot:5.1f}%")
print(f"total {tot/1e6:.2f} ms over {len(rows)} dispatches")
