# SQ counters of conv_win on a ts1-sized (or $1 rows, tensor stride $2) 96 -> 96 layer: two --pmc passes with --kernel-trace only
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/cp1 /tmp/cp2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/cp1 --output-format csv -- python $R/profiles/win_micro.py 5 ${1:-80000} ${2:-1} > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE -d /tmp/cp2 --output-format csv -- python $R/profiles/win_micro.py 5 ${1:-80000} ${2:-1} > /dev/null 2>&1
cd $R; for d in cp1 cp2; do f=$(ls -t $(find /tmp/$d -name "*counter_collection.csv") | head -1); python profiles/pmc_summary.py $f | grep -i -E "conv_win|kernel  "; done
