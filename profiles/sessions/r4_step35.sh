cd $GRAFT_REPO_ROOT
O=gpurun_out/r4s35; mkdir -p $O
for t in 256 512 1024 512 256; do
  touch canonicalvoting_amd/csrc/hv_decode.hip; CV_DEC_DEFS="-DDEC_SMALL_T=$t" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo -n "DEC_SMALL_T=$t :" >> $O/dec_small_t.txt
  for i in 1 2 3; do python3 bench.py --streams 1 --steps 80 --cpu-scenes 0 --train-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(' %.1f (decode %.3f)' % (d['value'], d['stage_ms_median']['decode']), end='')" >> $O/dec_small_t.txt; done
  echo >> $O/dec_small_t.txt
done
cat $O/dec_small_t.txt
