# round 5, step 6: what binds conv_win v4: timing ablations, each on a REAL rebuild of sparse_win.hip (the object is deleted first:
# build.py's staleness test does not see a changed -D), then SQ counters
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s6
mkdir -p $O
echo "ABL=0" > $O/ablate.txt
python profiles/win_micro.py 20 80000 1 >> $O/ablate.txt 2>&1
python profiles/win_micro.py 20 80000 2 >> $O/ablate.txt 2>&1
bash profiles/win_pmc.sh 80000 2 > $O/win_pmc_ts2.txt 2>&1
for abl in 2 8 10 4 1 5 15; do
  rm -f canonicalvoting_amd/_C/obj/sparse_win.hip.o
  CV_WIN_DEFS="-DCV_WIN_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "ABL=$abl" >> $O/ablate.txt
  python profiles/win_micro.py 20 80000 1 >> $O/ablate.txt 2>&1
  python profiles/win_micro.py 20 80000 2 >> $O/ablate.txt 2>&1
done
grep -v amdgpu.ids $O/ablate.txt; cat $O/win_pmc_ts2.txt
