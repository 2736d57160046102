#!/bin/bash
# stale-LDS hunt: the tile kernel with parts of its LDS poisoned at start; a changed result = an uninitialised read
for m in 0 1 2 4 8 16; do
  if [ $m = 0 ]; then CV_HV_DEFS="" python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1; else CV_HV_DEFS="-DHV_POISON=$m" python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1; fi
  echo "== HV_POISON=$m"; python profiles/vote_race_probe3.py 2>&1 | grep "alone\|saved"
done
python -m canonicalvoting_amd.csrc.build --force > /dev/null 2>&1
