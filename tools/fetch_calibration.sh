# dev tool (GPU box): what FETCH_SIZE reports for gemm_p9_kernel when every activation row has exactly ONE reader (N = 256: one
# column tile) and two (N = 512) - calibration of the counter on this kernel's LDS-DMA access pattern (MI355X_MICROARCH.md: "calibrate
# on a known byte count in your own access pattern")
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-fetch_cal}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LAB_CAL=1 LAB_NO_DIAG=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- $R/tools/build/gemm_lab > $O/lab.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_p" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            key = r["Kernel_Name"].split("::")[-1].split("(")[0] + " grid " + r.get("Grid_Size", "?")
            acc[key].append(float(r["Counter_Value"]))
print("M = 25216, K = 768: A = 38.73 MB (read once when N = 256), W = 0.39 MB per column tile and XCD")
for k, v in sorted(acc.items()):
    print("%-60s launches %3d  FETCH_SIZE raw %7.2f MB   x2 %7.2f MB" % (k, len(v), sum(v) / len(v) * 1024 / 1e6, 2 * sum(v) / len(v) * 1024 / 1e6))
PY
rm -rf $O/f
