# round 5, step 4: what binds conv_win v3: SQ counters and timing ablations (CV_WIN_ABL: 1 no window DMA, 2 no MFMA, 4 no weight DMA, 8 no fragment reads)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s4
mkdir -p $O
python profiles/win_micro.py 20 80000 1 > $O/ablate.txt 2>&1
python profiles/win_micro.py 20 80000 2 >> $O/ablate.txt 2>&1
bash profiles/win_pmc.sh 80000 2 > $O/win_pmc_ts2.txt 2>&1
bash profiles/win_pmc.sh 80000 1 > $O/win_pmc_ts1.txt 2>&1
for abl in 2 8 10 4 1 15; do
  CV_WIN_DEFS="-DCV_WIN_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "ABL=$abl" >> $O/ablate.txt
  python profiles/win_micro.py 20 80000 1 >> $O/ablate.txt 2>&1
  python profiles/win_micro.py 20 80000 2 >> $O/ablate.txt 2>&1
done
cat $O/ablate.txt; cat $O/win_pmc_ts2.txt $O/win_pmc_ts1.txt
