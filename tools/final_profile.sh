set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_under_rocprof.json 2> $O/stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic > $O/fetch.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic > $O/write.json 2> $O/write.err
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-overlap > $O/mfma.json 2> $O/mfma.err
echo "mfma done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp32 -- python3 $R/bench.py --mode postproc --batch 32 > $O/pp32.json 2> $O/pp32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp256 -- python3 $R/bench.py --mode postproc --batch 256 > $O/pp256.json 2> $O/pp256.err
echo "postproc done"
python3 $R/bench.py --mode postproc --batch 32 2>/dev/null | tail -1 > $O/pp32_line.json
python3 $R/bench.py --mode postproc --batch 256 2>/dev/null | tail -1 > $O/pp256_line.json
for m in "--mode train" "--mode train-yolo" "--models large --batch 64" "--models large --batch 64 --dtype mxfp8" "--no-split"; do
  python3 $R/bench.py $m --no-cpu-baseline 2>/dev/null | tail -1 >> $O/other_modes.jsonl
done
echo "all done"
