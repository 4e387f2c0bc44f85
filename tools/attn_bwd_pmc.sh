# dev tool (GPU box): PMC counters of the attention backward kernels (tools/attn_bwd_one.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-attn_bwd_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/a -- python3 $R/tools/attn_bwd_one.py > $O/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/b -- python3 $R/tools/attn_bwd_one.py > $O/b.log 2>&1 || exit 1
for k in attn_bwd_dkv attn_bwd_dq; do echo $k; python3 $R/tools/pmc_kernel_summary.py $O/a $k $O/b | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print('  ',k, round(v['avg_per_launch']), v['launches'])"; done
grep "attention backward" $O/a.log
rm -rf $O/a $O/b
