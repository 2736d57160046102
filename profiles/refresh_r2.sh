# regenerates the round-2 artefacts of profiles/r2/ in one gpurun call (copy gpurun_out/r2n/* to profiles/r2/ afterwards)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2n
mkdir -p $O
python bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/bench.json
python bench.py --steps 20 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_20steps.json
python bench.py --streams 1 --cpu-scenes 0 2>/dev/null | tail -1 > $O/bench_streams1.json
python bench.py --streams 1 --cpu-scenes 0 --predictions network 2>/dev/null | tail -1 > $O/bench_streams1_network_predictions.json
python bench.py --streams 1 --cpu-scenes 0 --large --points 300000 --steps 40 2>/dev/null | tail -1 > $O/bench_streams1_300k.json
python bench.py --mode train --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_train.json
python bench.py --mode train --steps 10 --warmup 2 --dtype bf16 2>/dev/null | tail -1 > $O/bench_train_bf16.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 10 --warmup 3 --cpu-scenes 0 > /tmp/p1.log 2>&1; f=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats.csv)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --cpu-scenes 0 > /tmp/p3.log 2>&1; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_streams6.csv)
bash profiles/trace_one.sh r2n > /dev/null 2>&1
MICRO_HL=1 bash profiles/conv_pmc.sh > $O/conv_pmc_hl.txt 2>&1
bash profiles/conv_l2_pmc.sh > $O/conv_l2_pmc.txt 2>&1
bash profiles/vote_pmc.sh > $O/vote_pmc.log 2>&1; cp gpurun_out/vote_pmc/* $O/
for s in 4 6 8 10; do
  echo "streams=$s" >> $O/sweep_streams.txt
  python bench.py --steps 240 --cpu-scenes 0 --streams $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['stage_ms'], round(d['roofline']['frac'],4))" >> $O/sweep_streams.txt
done
ls -la $O
