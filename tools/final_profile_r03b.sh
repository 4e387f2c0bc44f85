# round 3, part B (GPU box): training profiles, detector stage, isolated kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
bash $R/tools/train_profile.sh r03f/train > $O/train_profile.log 2>&1; echo "train done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/det -- python3 $R/tools/detect_only.py > $O/det.log 2> $O/det.err; echo "det done"
python3 $R/tools/stage_split.py > $O/stage_split.txt 2>&1; echo "stage done"
python3 $R/tools/conv_layers.py > $O/conv_layers.txt 2>&1; echo "conv done"
timeout -k 10 300 $R/tools/build/gemm_lab > $O/gemm_lab.txt 2>&1; echo "lab done"
python3 $R/tools/attn_ablate.py > $O/attn_ablate.txt 2>&1; ATTN_R=64 python3 $R/tools/attn_ablate.py >> $O/attn_ablate.txt 2>&1; echo "attn done"
bash $R/tools/attn_pmc.sh r03f/attn_pmc > $O/attn_pmc.log 2>&1; echo "attn pmc done"
bash $R/tools/traffic_by_instance.sh r03f/traffic > $O/traffic_by_instance.txt 2>&1; echo "traffic done"
python3 $R/tools/fp8_bench.py > $O/fp8_bench.txt 2>&1; echo "fp8 done"
python3 $R/tools/cus_sweep.py > $O/cus_sweep.txt 2>&1; echo "cus done"
python3 $R/tools/c2f_bench.py > $O/c2f_bench.txt 2>&1; echo "c2f done"
bash $R/tools/c2f_pmc.sh r03f/c2f_pmc32 > $O/c2f_pmc32.log 2>&1; C2F_C=16 C2F_N=1 C2F_H=160 bash $R/tools/c2f_pmc.sh r03f/c2f_pmc16 > $O/c2f_pmc16.log 2>&1; echo "c2f pmc done"
C2F_STAMPS=1 python3 $R/tools/c2f_one.py > $O/c2f_stamps.txt 2>&1; echo "c2f stamps done"
python3 $R/tools/wgrad_bench.py > $O/wgrad_bench.txt 2>&1; echo "wgrad done"
LAB_M=6304 LAB_TRAIN=1 LAB_R1=1 timeout -k 10 300 $R/tools/build/gemm_lab > $O/gemm_lab_m6304.txt 2>&1; echo "lab 6304 done"
E2E_C2F=1 python3 $R/tools/e2e_ab.py > $O/e2e_c2f.txt 2>&1; echo "e2e c2f done"
python3 $R/tools/crop_bench.py > $O/crop_bench.txt 2>&1; echo "crop done"
bash $R/tools/crop_pmc.sh r03f/crop_pmc > $O/crop_pmc.log 2>&1; bash $R/tools/nms_pmc.sh r03f/nms_pmc > $O/nms_pmc.log 2>&1; echo "crop / nms pmc done"
python3 $R/tools/write_ceiling.py > $O/write_ceiling.txt 2>&1
for b in 8 32 128 256 512 1024; do python3 $R/bench.py --mode postproc --batch $b 2>/dev/null | tail -1; done > $O/pp_sweep.jsonl; echo "sweep done"
