# round 3, part A (GPU box): headline bench line, rocprofv3 summaries of the bench, PMC passes, post-processing, other modes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-overlap > $O/bench_single_under_rocprof.json 2> $O/stats_single.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic > $O/write.json 2> $O/write.err
echo "traffic done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-overlap > $O/mfma.json 2> $O/mfma.err
echo "mfma done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp32 -- python3 $R/bench.py --mode postproc --batch 32 > $O/pp32.json 2> $O/pp32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pp256 -- python3 $R/bench.py --mode postproc --batch 256 > $O/pp256.json 2> $O/pp256.err
python3 $R/bench.py --mode postproc --batch 32 2>/dev/null | tail -1 > $O/pp32_line.json
python3 $R/bench.py --mode postproc --batch 256 2>/dev/null | tail -1 > $O/pp256_line.json
echo "postproc done"
rm -f $O/other_modes.jsonl
for m in "--mode train" "--mode train --batch 128" "--mode train-yolo" "--models large --batch 64" "--models large --batch 64 --dtype mxfp8" "--no-split" "--no-overlap"; do
  python3 $R/bench.py $m --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 >> $O/other_modes.jsonl
done
echo "all done"
