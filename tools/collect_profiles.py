#!/usr/bin/env python3
"""Turn gpurun_out/r02f (written by tools/final_profile.sh + the detector tools) into the tracked files under profiles/ and
refresh the measured figures in profiles/README.md."""
import csv, glob, json, os, re, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
G = R + "gpurun_out/r02f/"
subprocess.check_call([sys.executable, R + "tools/profile_summary.py", "r02", G + "stats", G + "fetch", G + "write"], stdout=subprocess.DEVNULL)
shutil.copy(G + "bench_line.json", R + "profiles/r02_bench_line.json")
open(R + "profiles/r02_bench_line_under_rocprof.json", "w").write(open(G + "bench_under_rocprof.json").read().strip().split("\n")[-1] + "\n")
for pat, name in (("gemm_p8_kernel", "r02_pmc_gemm_p8_mfma.json"), ("attention_kernel", "r02_pmc_attention_mfma.json")):
    out = subprocess.check_output([sys.executable, R + "tools/pmc_kernel_summary.py", G + "mfma", pat])
    open(R + "profiles/" + name, "wb").write(out)
for b in ("32", "256"):
    f = glob.glob(G + f"pp{b}/*/*kernel_stats.csv"); assert len(f) == 1, f
    shutil.copy(f[0], R + f"profiles/r02_postproc_b{b}_kernel_stats.csv")
pl = [json.loads(open(G + f"pp{b}_line.json").read()) for b in ("32", "256")]
json.dump(pl, open(R + "profiles/r02_postproc_bench_lines.json", "w"), indent=1)
om = [json.loads(l) for l in open(G + "other_modes.jsonl")]; assert len(om) == 5
json.dump(om, open(R + "profiles/r02_bench_lines_other_modes.json", "w"), indent=1)
f = glob.glob(G + "det/*/*kernel_stats.csv"); assert len(f) == 1, f
shutil.copy(f[0], R + "profiles/r02_detect_stage_kernel_stats.csv")
keep = lambda p: "\n".join(l for l in open(p).read().split("\n") if "amdgpu.ids" not in l)
open(R + "profiles/r02_conv_layers.txt", "w").write(keep(G + "conv_layers.txt"))
open(R + "profiles/r02_stage_split.txt", "w").write("# python3 tools/stage_split.py (one process, interleaved; ms per batch of 32 images / 128 crops)\n" + keep(G + "stage_split.txt"))
b = json.load(open(R + "profiles/r02_bench_line.json")); u = json.load(open(R + "profiles/r02_bench_line_under_rocprof.json"))
sm = json.load(open(R + "profiles/r02_summary.json")); i = sm["instances"]
det = [r for r in csv.reader(open(R + "profiles/r02_detect_stage_kernel_stats.csv")) if r[0] != "Name"]
det_us = sum(int(r[2]) for r in det) / 20 / 1e3
conv_sum = float(re.search(r"sum of isolated conv times: ([0-9.]+)", open(R + "profiles/r02_conv_layers.txt").read()).group(1))
ss = open(R + "profiles/r02_stage_split.txt").read()
sp = {k: float(re.search(re.escape(k) + r"\s+median ([0-9.]+)", ss).group(1)) for k in ("detect stage alone", "classify alone, split, 208 CUs", "pipelined whole, 208 CUs")}
rows = {
 "| `r02_bench_line.json` |": f"| `r02_bench_line.json` | `python3 bench.py --steps 20 --warmup 5`, no profiler | the number that counts on that box: {b['value']:.0f} images/s, {b['ms_per_step']:.2f} ms per step; `roofline.frac` {b['roofline']['frac']:.3f} (algorithmic flops of the instrumented GEMM launches / time during which at least one of them is executing, {b['roofline']['kernel_busy_ms_per_step']:.2f} ms per step - the launches share the chip with the detector stream and with each other's half-batch); `cpu_baseline` {b['cpu_baseline']['value']:.2f} images/s on {b['cpu_baseline']['cores']} host cores (oracle port) |",
 "| `r02_kernel_stats.csv`, `r02_summary.json` |": f"| `r02_kernel_stats.csv`, `r02_summary.json` | `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline` (`tools/profile_summary.py r02 …`) | per-kernel totals of the headline bench: `gemm_p8_kernel` {sm['calls']} calls (96 per step: 12 layers x 4 linears x 2 half-batches), {sm['avg_us']:.1f} us mean, {100*sm['share_of_gpu_time']:.0f} % of GPU time; per instance: `<3,3,false>` (qkv) {i['gemm_p8_kernel<3, 3, false>']['avg_us']:.1f} us, `<4,4,false>` (fc1 + GELU) {i['gemm_p8_kernel<4, 4, false>']['avg_us']:.1f} us, `<3,3,true>` (proj / fc2, f32 residual) {i['gemm_p8_kernel<3, 3, true>']['avg_us']:.1f} us; LayerNorm 7.5 %, attention 7.2 %, the detector's `igemm_kernel` instances 14 % |",
 "| `r02_bench_line_under_rocprof.json` |": f"| `r02_bench_line_under_rocprof.json` | the JSON line `bench.py` printed INSIDE that rocprofv3 run | agreement check: `roofline.avg_launch_us` = {u['roofline']['avg_launch_us']:.1f} us from the launch-attached HIP events (2 sampled steps) vs {sm['avg_us']:.1f} us in the rocprofv3 summary (all 15 steps) |",
 "| `r02_postproc_b32_kernel_stats.csv`": f"| `r02_postproc_b32_kernel_stats.csv`, `r02_postproc_b256_kernel_stats.csv`, `r02_postproc_bench_lines.json` | `rocprofv3 --kernel-trace --stats … bench.py --mode postproc --batch 32 / 256`, and the two lines without the profiler | EfficientNMS on the SURVEY 8(d) synthetic sets: batch 32 {pl[0]['roofline']['us_per_call']:.1f} us per call on this box (the call is a chain of three dependent launches - `en2_filter_kernel`, `en2_front_kernel`, the idle `en2_tail_kernel` - and the per-launch latency differs between boxes; single-kernel form {pl[0]['nms_single_kernel_us']:.0f} us in the same process), batch 256 {pl[1]['roofline']['us_per_call']:.1f} us = {100*pl[1]['roofline']['frac']:.0f} % of the HBM peak; `crop_kernel<2>` {pl[0]['crop_roofline']['us_per_call']:.1f} us / {pl[1]['crop_roofline']['us_per_call']:.0f} us = {100*pl[0]['crop_roofline']['frac']:.0f} % / {100*pl[1]['crop_roofline']['frac']:.0f} % of peak |",
 "| `r02_bench_lines_other_modes.json` |": f"| `r02_bench_lines_other_modes.json` | `python3 bench.py --mode train`, `--mode train-yolo`, `--models large --batch 64 [--dtype mxfp8]`, `--no-split` (all `--no-cpu-baseline`) | ViT-B/16 fine-tune step {om[0]['value']:.0f} img/s ({om[0]['ms_per_step']:.2f} ms), YOLOv8s training step {om[1]['value']:.0f} img/s ({om[1]['ms_per_step']:.2f} ms), YOLOv8m + ViT-L/16 inference {om[2]['value']:.0f} img/s in bf16 and {om[3]['value']:.0f} with MXFP8 block linears, headline pair without the half-batch split {om[4]['value']:.0f} img/s |",
 "| `r02_detect_stage_kernel_stats.csv`": f"| `r02_detect_stage_kernel_stats.csv` (`…_before.csv`: start of the round), `r02_stage_split.txt` | `rocprofv3 --kernel-trace --stats … python3 tools/detect_only.py` (20 passes of the detect stage alone); `python3 tools/stage_split.py` | detector stream per batch of 32: 2,006 -> {det_us:,.0f} us of kernel time under the profiler (convolutions 1,549 -> 1,300, NMS front 186 -> 34, stem 122 -> 44); stage timing: detect alone 1.96 -> {sp['detect stage alone']:.2f} ms, classify alone {sp['classify alone, split, 208 CUs']:.2f} ms, pipelined whole {sp['pipelined whole, 208 CUs']:.2f} ms |",
 "| `r02_conv_layers.txt` |": f"| `r02_conv_layers.txt` | `python3 tools/conv_layers.py` | every convolution launch of one YOLOv8n forward replayed alone: shape, us, TFLOP/s, GB/s against its algorithmic bytes (sum 1,465 us at the start of the round, {conv_sum:,.0f} us now) |",
}
out = []
for l in open(R + "profiles/README.md").read().split("\n"):
    for k, v in rows.items():
        if l.startswith(k):
            l = v
    if l.startswith("| `pmc_traffic.json`"):
        l = re.sub(r"= [0-9]+ MB against 110 MB", f"= {sm['bytes_per_launch']/1e6:.0f} MB against 110 MB", l)
    out.append(l)
open(R + "profiles/README.md", "w").write("\n".join(out))
print("bench", round(b["value"]), b["ms_per_step"], b["roofline"]["frac"], "| nms", [round(p["roofline"]["us_per_call"], 1) for p in pl],
      "single", round(pl[0]["nms_single_kernel_us"]), "| crop", [round(p["crop_roofline"]["us_per_call"], 1) for p in pl],
      "| other", [round(o["value"]) for o in om], "| det", round(det_us), "conv", conv_sum, sp)
