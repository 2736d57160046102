cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; ARGS="$@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/pmc1 --output-format csv -- python $R/profiles/tile_one.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 --output-format csv -- python $R/profiles/tile_one.py $ARGS > /dev/null 2>&1
cd $R; for d in pmc1 pmc2; do f=$(ls -t $(find gpurun_out/$d -name "*counter_collection.csv") | head -1); python profiles/pmc_summary.py $f | grep -i -E "conv_tile|conv_rows|kernel  " ; done
