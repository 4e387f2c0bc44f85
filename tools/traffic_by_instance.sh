# dev tool (GPU box): FETCH_SIZE / WRITE_SIZE per launch of every gemm_p9_kernel instance of the headline bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-traffic}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic $2 > $O/$c.json 2> $O/$c.err || exit 1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p9_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                key = r["Kernel_Name"].split("gemm_p9_kernel")[1].split("(")[0] + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
                acc[key][c].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1); w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print("%-40s launches %4d  fetch x2 %7.1f MB  write %7.1f MB  total %7.1f MB" % (k, len(v["FETCH_SIZE"]), 2 * f * 1024 / 1e6, w * 1024 / 1e6, (2 * f + w) * 1024 / 1e6))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
