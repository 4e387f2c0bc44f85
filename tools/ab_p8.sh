# end-to-end A/B (dev tool; run on the GPU box): tile schedule of the persistent GEMM
for rd in 1 2; do
for opt in "linear_p8_sched=0" "linear_p8_sched=1"; do
    echo "== round $rd $opt"
    YV_OPTIONS=$opt timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
done
