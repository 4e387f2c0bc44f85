# dev tool (GPU box): PMC counters of en2_front_kernel at 256 images (bench.py --mode postproc --batch 256)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-nms_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/a -- python3 $R/bench.py --mode postproc --batch 256 --steps 5 > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/b -- python3 $R/bench.py --mode postproc --batch 256 --steps 5 > $O/b.log 2>&1 || exit 1
python3 $R/tools/pmc_kernel_summary.py $O/a en2_front $O/b > $O/summary.json
cat $O/summary.json
rm -rf $O/a $O/b
