# SQ / LDS counters of the vote tile kernel (DESIGN 4.1 called it LDS-pipe bound without counters).
# Separate rocprofv3 --pmc passes (with --kernel-trace only), a few counters each; per launch of hv_fwd_tiles.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --streams 1 --stage vote_decode --steps 10 --warmup 2 --cpu-scenes 0 --min-warm-seconds 0 --measure-traffic 0"
O=$R/gpurun_out/${1:-vote_pmc_sq}; mkdir -p $O; : > $O/vote_pmc_sq.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1)); rm -rf /tmp/vsq$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/vsq$i --output-format csv -- $CMD > /tmp/vsq$i.log 2>&1
  f=$(find /tmp/vsq$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then echo "## pass $i: $set" >> $O/vote_pmc_sq.txt; python $R/profiles/pmc_summary.py $f | grep -i "kernel\|hv_fwd_tiles\|hv_prep" >> $O/vote_pmc_sq.txt; else echo "## pass $i: $set FAILED" >> $O/vote_pmc_sq.txt; tail -3 /tmp/vsq$i.log >> $O/vote_pmc_sq.txt; fi
done
cat $O/vote_pmc_sq.txt
