# round 5, step 26: kernel time of the training step (3 x 80k), per kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5s26
mkdir -p $O
(cd /tmp && rm -rf /tmp/pt && CV_TRAIN_FWD_HL=${HL:-1} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 8 --warmup 2 --cpu-scenes 0 > /tmp/pt.log 2>&1; f=$(find /tmp/pt -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv; tail -1 /tmp/pt.log > $O/bench_train_rocprof.json)
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5s26/train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6)
for r in rows[:28]:
    print('%-70s %6s calls %9.3f ms %5.1f %%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['Percentage'])))
P
