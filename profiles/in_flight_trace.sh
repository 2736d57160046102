# pass T of in_flight_counters.sh alone, per hardware-queue count (the PMC directories of an earlier run are reused if present)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
for q in ${QUEUES:-4 8}; do
  rm -rf /tmp/ifT$q
  GPU_MAX_HW_QUEUES=$q timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ifT$q -- python $R/bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 > $O/if_trace_bench_q$q.json 2> /tmp/ifT$q.err
  python $R/profiles/in_flight_summary.py /tmp/ifT$q - - - $O/if_trace_bench_q$q.json > $O/in_flight_trace_q$q.txt 2>&1; gzip -c $(find /tmp/ifT$q -name "*kernel_trace.csv" | head -1) > $O/kernel_trace_q$q.csv.gz
  head -24 $O/in_flight_trace_q$q.txt
  GPU_MAX_HW_QUEUES=$q python $R/bench.py --steps 240 --cpu-scenes 0 --train-steps 0 --measure-traffic 0 2>/dev/null | tail -1 > $O/bench_q$q.json
  python -c "import json; d=json.load(open('$O/bench_q$q.json')); print('GPU_MAX_HW_QUEUES=$q without the tracer: %.1f scenes/s' % d['value'])"
done
