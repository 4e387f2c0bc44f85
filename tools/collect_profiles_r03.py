#!/usr/bin/env python3
"""Turn gpurun_out/r03f (tools/final_profile_r03a.sh + final_profile_r03b.sh, one MI355X box each) into the tracked files under
profiles/ (r03_*, pmc_traffic.json)."""
import csv, glob, json, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
G = R + "gpurun_out/r03f/"
P = R + "profiles/"
keep = lambda p: "\n".join(l for l in open(p).read().split("\n") if "amdgpu.ids" not in l)
last = lambda p: open(p).read().strip().split("\n")[-1] + "\n"
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: a re-run leaves the older run's rocprofv3 directory beside the new one"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    assert fs, pattern
    return [fs[-1]]
subprocess.check_call([sys.executable, R + "tools/profile_summary.py", "r03", G + "stats", G + "fetch", G + "write"], stdout=subprocess.DEVNULL)
open(P + "r03_bench_line.json", "w").write(last(G + "bench_line.json"))
open(P + "r03_bench_line_under_rocprof.json", "w").write(last(G + "bench_under_rocprof.json"))
f = newest(G + "stats_single/*/*kernel_stats.csv")
open(P + "r03_kernel_stats_single_stream.csv", "w").write(
    "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-overlap\n" + open(f[0]).read())
for pat, name in (("gemm_p9_kernel", "r03_pmc_gemm_p9_mfma.json"), ("attention_pipe_kernel", "r03_pmc_attention_in_bench.json")):
    open(P + name, "wb").write(subprocess.check_output([sys.executable, R + "tools/pmc_kernel_summary.py", G + "mfma", pat]))
shutil.copy(G + "attn_pmc/summary.json", P + "r03_pmc_attention_pipe.json")
shutil.copy(G + "c2f_pmc32/summary.json", P + "r03_pmc_c2f_model4.json")
shutil.copy(G + "crop_pmc/summary.json", P + "r03_pmc_crop.json")
shutil.copy(G + "nms_pmc/summary.json", P + "r03_pmc_nms_front.json")
with open(P + "r03_postproc_batch_sweep.txt", "w") as f_:
    f_.write("# for b in 8 32 128 256 512 1024: python3 bench.py --mode postproc --batch b  (EfficientNMS call = en2_filter + en2_front + en2_tail; crop_kernel<2> at 4 crops per image)\n")
    for l_ in open(G + "pp_sweep.jsonl"):
        d_ = json.loads(l_)
        f_.write("%5d images: NMS %6.1f us %5.1f %% of 8 TB/s | crop (4 per image) %6.1f us %5.1f %%\n" % (d_["config"]["batch"], d_["roofline"]["us_per_call"], d_["roofline"]["frac"] * 100, d_["crop_roofline"]["us_per_call"], d_["crop_roofline"]["frac"] * 100))
shutil.copy(G + "c2f_pmc16/summary.json", P + "r03_pmc_c2f_model2.json")
for b in ("32", "256"):
    f = newest(G + f"pp{b}/*/*kernel_stats.csv")
    shutil.copy(f[0], P + f"r03_postproc_b{b}_kernel_stats.csv")
json.dump([json.loads(open(G + f"pp{b}_line.json").read()) for b in ("32", "256")], open(P + "r03_postproc_bench_lines.json", "w"), indent=1)
json.dump([json.loads(l) for l in open(G + "other_modes.jsonl")], open(P + "r03_bench_lines_other_modes.json", "w"), indent=1)
for n in ("train_vit", "train_yolo"):
    open(P + f"r03_{n}_kernel_stats.csv", "w").write(
        f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --mode {'train' if n == 'train_vit' else 'train-yolo'} --steps 10 --warmup 3 --no-cpu-baseline\n"
        + open(G + f"train/{n}_kernel_stats.csv").read())
json.dump({n: json.loads(last(G + f"train/{n}_line.json")) for n in ("train_vit", "train_yolo")}, open(P + "r03_train_bench_lines.json", "w"), indent=1)
f = newest(G + "det/*/*kernel_stats.csv")
shutil.copy(f[0], P + "r03_detect_stage_kernel_stats.csv")
for src, dst, head in (("stage_split.txt", "r03_stage_split.txt", "# python3 tools/stage_split.py (one process, interleaved; ms per batch of 32 images / 128 crops)\n"),
                       ("conv_layers.txt", "r03_conv_layers.txt", ""), ("gemm_lab.txt", "r03_gemm_lab.txt", "# tools/build/gemm_lab (stand-alone, interleaved in one process; p8 = round-2 kernel, p9 = shipped)\n"),
                       ("attn_ablate.txt", "r03_attention_ablate.txt", "# python3 tools/attn_ablate.py (ablate 0 = shipped pipelined kernel, 5 = round-2 kernel, 11 / 12 / 13 = without fetches / compute / stores)\n"),
                       ("traffic_by_instance.txt", "r03_traffic_by_instance.txt", "# bash tools/traffic_by_instance.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py, per gemm_p9_kernel instance)\n"),
                       ("fp8_bench.txt", "r03_fp8_gemm_bench.txt", "# python3 tools/fp8_bench.py (ViT-L shapes at 50,432 rows, epilogues included)\n"),
                       ("cus_sweep.txt", "r03_cus_sweep.txt", "# python3 tools/cus_sweep.py (pipelined step vs the CU budget of the persistent GEMMs)\n"),
                       ("c2f_bench.txt", "r03_c2f_bench.txt", "# python3 tools/c2f_bench.py (backbone C2f blocks of YOLOv8n at batch 32: yv_c2f_fused vs yv_conv2d x 4 / 6)\n"),
                       ("c2f_stamps.txt", "r03_c2f_stamps.txt", "# C2F_STAMPS=1 python3 tools/c2f_one.py (model.4: cycle stamps of wave 0 at the phase boundaries)\n"),
                       ("wgrad_bench.txt", "r03_wgrad_bench.txt", "# python3 tools/wgrad_bench.py (weight-gradient GEMMs of the ViT-B/16 fine-tune step by number of token slices; 0 = shipped)\n"),
                       ("gemm_lab_m6304.txt", "r03_gemm_lab_m6304.txt", "# LAB_M=6304 LAB_TRAIN=1 LAB_R1=1 tools/build/gemm_lab (the trainer's shapes: 32 images = 6,304 rows)\n"),
                       ("crop_bench.txt", "r03_crop_bench.txt", "# python3 tools/crop_bench.py (crop_kernel<2> by row groups per block; 0 = the launcher's choice; 'wide' = 340-640 pixel boxes)\n"),
                       ("write_ceiling.txt", "r03_write_ceiling.txt", "# python3 tools/write_ceiling.py (plain torch fill / copy of the crop kernel's 308 MB output: what a streaming kernel reaches on this box)\n"),
                       ("e2e_c2f.txt", "r03_e2e_c2f_ab.txt", "# E2E_C2F=1 python3 tools/e2e_ab.py (whole pipelined step, backbone C2f blocks layer by layer vs fused, interleaved in one process)\n")):
    open(P + dst, "w").write(head + keep(G + src))
print("ok")
