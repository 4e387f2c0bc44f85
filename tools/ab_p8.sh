# end-to-end A/B (dev tool; run on the GPU box): persistent-GEMM grid size under the three-stream schedule
for opt in "linear_p8=2,linear_p8_cus=208" "linear_p8=2,linear_p8_cus=192" "linear_p8=2,linear_p8_cus=176" "linear_p8=2,linear_p8_cus=160" "linear_p8=2,linear_p8_cus=144" "linear_p8=2,linear_p8_cus=200"; do
  for sp in ""; do
    echo "== $opt $sp"
    YV_OPTIONS=$opt timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $sp 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
  done
done
