# round 5, step 8: what the MFMA-only configuration of conv_win v5 is made of (real rebuilds)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s8
mkdir -p $O
: > $O/ablate.txt
for abl in 13 29 45 61 125 32 48; do
  rm -f canonicalvoting_amd/_C/obj/sparse_win.hip.o
  CV_WIN_DEFS="-DCV_WIN_ABL=$abl" python -m canonicalvoting_amd.csrc.build > /dev/null 2>&1
  echo "ABL=$abl" >> $O/ablate.txt
  python profiles/win_micro.py 20 80000 2 >> $O/ablate.txt 2>&1
done
grep -v amdgpu.ids $O/ablate.txt
