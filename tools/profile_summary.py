#!/usr/bin/env python3
"""Condense rocprofv3 outputs (gpurun_out/<dir>) into the tracked summaries under profiles/.
usage: profile_summary.py <round-tag> <stats_dir> [<fetch_pmc_dir> <write_pmc_dir>] [--kernel PATTERN]
PATTERN (default gemm_p9_kernel) selects the dominant kernel's rows (all template instances together)."""
import csv, glob, json, os, sys

argv = list(sys.argv[1:])
pat = "gemm_p9_kernel"
if "--kernel" in argv:
    i = argv.index("--kernel"); pat = argv[i + 1]; del argv[i:i + 2]
tag, stats_dir = argv[0], argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
f = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline\n")
    w = csv.DictWriter(o, fieldnames=list(rows[0].keys())); w.writeheader(); [w.writerow(r) for r in rows]
g = [r for r in rows if pat in r["Name"]]
calls = sum(int(r["Calls"]) for r in g); tot = sum(float(r["TotalDurationNs"]) for r in g)
summary = {"kernel": pat, "calls": calls, "avg_us": tot / calls / 1e3,
           "share_of_gpu_time": tot / sum(float(r["TotalDurationNs"]) for r in rows),
           "instances": {r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""):
                         {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3} for r in g}}
if len(argv) >= 4:
    def pmc(d, name):
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"] and r["Counter_Name"] == name]
        return sum(vals) / len(vals), len(vals)
    fetch_kb, n1 = pmc(argv[2], "FETCH_SIZE")
    write_kb, n2 = pmc(argv[3], "WRITE_SIZE")
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced read stream (double it); WRITE_SIZE
    # is exact for 16-byte stores.  Units: KB per dispatch.
    summary.update({"fetch_size_kb_raw_avg": fetch_kb, "write_size_kb_avg": write_kb, "pmc_dispatches": [n1, n2],
                    "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0})
    json.dump({"kernel": pat, "round": tag, "bytes_per_launch": summary["bytes_per_launch"], "launches_profiled": [n1, n2],
               "source": f"profiles/{tag}_summary.json"}, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
