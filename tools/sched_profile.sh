# dev tool (GPU box): per-kernel totals of the headline bench under the three schedules
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-sched}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in "split:" "nosplit:--no-split" "single:--no-overlap"; do
  n=${m%%:*}; f=${m#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic $f > $O/$n.json 2> $O/$n.err || exit 1
  cp $(ls $O/$n/*/*kernel_stats.csv | head -1) $O/${n}_kernel_stats.csv
  rm -rf $O/$n
  echo "$n done"
done
