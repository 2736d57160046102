# the bank-aware drain (-DHV_BANK_BUCKET=1) against the default, same box, same session:
# kernel time (rocprofv3 --kernel-trace --stats of profiles/vote_time.py), the LDS conflict counters, exactness, scene rate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O; : > $O/vote_bank_bucket.txt
for v in default bucket default bucket; do
  if [ $v = bucket ]; then export CV_HV_DEFS="-DHV_BANK_BUCKET=1"; else unset CV_HV_DEFS; fi
  python -m canonicalvoting_amd.csrc.build > /dev/null || exit 1
  echo "## $v (CV_HV_DEFS='$CV_HV_DEFS')" >> $O/vote_bank_bucket.txt
  (cd /tmp && rm -rf /tmp/vb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vb -- python $GRAFT_REPO_ROOT/profiles/vote_time.py > /tmp/vb.log 2>&1; grep "event ms" /tmp/vb.log; f=$(find /tmp/vb -name "*kernel_stats.csv" | head -1); grep -E "hv_fwd_tiles" $f | cut -d, -f1-5) >> $O/vote_bank_bucket.txt
  (cd /tmp && rm -rf /tmp/vc && rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d /tmp/vc --output-format csv -- python $GRAFT_REPO_ROOT/profiles/vote_time.py > /tmp/vc.log 2>&1; f=$(find /tmp/vc -name "*counter_collection.csv" | head -1); python $GRAFT_REPO_ROOT/profiles/pmc_summary.py $f | grep -i "kernel \|hv_fwd_tiles") >> $O/vote_bank_bucket.txt
  python -m pytest tests/test_vote_gpu.py -q -m gpu -x 2>&1 | tail -1 >> $O/vote_bank_bucket.txt
  python bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench 240 steps: %.1f scenes/s, hv_fwd_tiles in region %.3f ms (frac %.3f), isolated %.3f ms (frac %.3f)' % (d['value'], d['roofline']['avg_ms'], d['roofline']['frac'], d['roofline']['isolated_avg_ms'], d['roofline']['isolated_frac']))" >> $O/vote_bank_bucket.txt
done
unset CV_HV_DEFS; python -m canonicalvoting_amd.csrc.build > /dev/null
cat $O/vote_bank_bucket.txt
