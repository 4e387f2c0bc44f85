#!/usr/bin/env python3
"""Dev tool: average PMC counter values per launch for kernels whose name contains a pattern.
usage: pmc_kernel_summary.py <rocprofv3 out dir> <name pattern> [more dirs...]   (prints JSON)"""
import collections, csv, glob, json, os, sys
pat = sys.argv[2]
out = {}
for d in [sys.argv[1]] + sys.argv[3:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, n = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc:
            out[k] = {"avg_per_launch": acc[k] / n[k], "launches": n[k]}
print(json.dumps(out, indent=1))
