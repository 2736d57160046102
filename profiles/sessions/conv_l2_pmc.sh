# L2 hit rate and memory-side traffic of the mask-sorted ts1 conv (80k rows, 96 -> 96, hl format): separate --pmc passes
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
export MICRO_HL=${MICRO_HL:-1}
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/cl2
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/cl2 --output-format csv -- python $R/profiles/conv_micro.py 5 > /dev/null 2>&1
  f=$(ls -t $(find /tmp/cl2 -name "*counter_collection.csv") | head -1)
  python $R/profiles/pmc_summary.py $f | grep -i -E "conv_hl|conv_rows|conv_finish|kernel  "
done
