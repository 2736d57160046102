#!/bin/bash
# round 3 step 3: vote work lists - parity, timing (80k / 300k), tick profile, bench
O=gpurun_out/r3d; mkdir -p $O
python -m pytest tests/test_vote_gpu.py tests/test_production_size_gpu.py tests/test_decode_gpu.py tests/test_proposals_gpu.py -m gpu -x -q > $O/pytest_vote.log 2>&1; tail -3 $O/pytest_vote.log
for l in 1 0; do
  echo "CV_HV_LISTS=$l" >> $O/vote_time.txt
  CV_HV_LISTS=$l python profiles/vote_time.py >> $O/vote_time.txt 2>&1
  CV_HV_LISTS=$l python profiles/vote_time.py --large >> $O/vote_time.txt 2>&1
done
python profiles/vote_time.py --ticks >> $O/vote_time.txt 2>&1
grep -v amdgpu.ids $O/vote_time.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scenes 0 > $O/b20.json 2> $O/err.txt
python3 bench.py --gpus 1 --steps 240 --warmup 5 --cpu-scenes 0 > $O/b240.json 2>> $O/err.txt
tail -c 300 $O/err.txt
