# counters of the weight-gradient kernels inside a training step (dispatch counters: the profiler serialises the kernels)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; : > $O/wgrad_pmc.txt
CMD="python $R/bench.py --mode train --steps 3 --warmup 2 --measure-traffic 0"
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf /tmp/wg$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/wg$i --output-format csv -- $CMD > /tmp/wg$i.log 2>&1
  f=$(find /tmp/wg$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "## pass $i: $set" >> $O/wgrad_pmc.txt; python $R/profiles/pmc_summary.py $f | grep -i "kernel \|conv_wgrad\|wgrad_reduce\|conv_hd\|conv_hl\|bn_col_reduce4\|bn_backward" >> $O/wgrad_pmc.txt; else echo "## pass $i: $set FAILED" >> $O/wgrad_pmc.txt; tail -3 /tmp/wg$i.log >> $O/wgrad_pmc.txt; fi
done
cat $O/wgrad_pmc.txt
