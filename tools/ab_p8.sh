# end-to-end A/B (dev tool; run on the GPU box): persistent GEMM on/off, grid size, schedules
for opt in "linear_p8=0" "linear_p8=2,linear_p8_cus=256" "linear_p8=2,linear_p8_cus=224" "linear_p8=2,linear_p8_cus=208"; do
  for sp in "" "--no-split"; do
    echo "== $opt $sp"
    YV_OPTIONS=$opt timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $sp 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
  done
done
