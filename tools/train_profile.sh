# dev tool (GPU box): per-kernel totals of the two training bench modes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-train_prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in "train_vit:--mode train" "train_yolo:--mode train-yolo"; do
  n=${m%%:*}; f=${m#*:}
  python3 $R/bench.py $f --steps 10 --warmup 3 --no-cpu-baseline > $O/${n}_line.json 2> $O/${n}_line.err || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python3 $R/bench.py $f --steps 10 --warmup 3 --no-cpu-baseline > $O/${n}_under_rocprof.json 2> $O/$n.err || exit 1
  cp $(ls $O/$n/*/*kernel_stats.csv | head -1) $O/${n}_kernel_stats.csv
  rm -rf $O/$n
  echo "$n done"
done
