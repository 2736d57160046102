# last validation of round 3 on the final tree: GPU suite, smoke, the driver's command, the default 240-step line, rocprof of the driver-shaped run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3final3; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 6 --cpu-scenes 0 --train-steps 0 > $O/bench_under_rocprof_default_streams.json 2>/dev/null; f=$(find /tmp/p3 -name "*kernel_stats.csv" | head -1); cp "$f" $O/full_path_kernel_stats_default_streams.csv)
python3 bench.py --steps 240 2>/dev/null | tail -1 > $O/bench.json
for f in bench_driver_cmd bench_under_rocprof_default_streams bench; do python3 -c "
import json
r=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); ro=r['roofline']
print('$f', round(r['value'],1), r['steps'], 'frac', round(ro['frac'],3), 'avg_ms', round(ro['avg_ms'],3), 'iso', ro['isolated_frac'] and round(ro['isolated_frac'],3), r['stage_ms_isolated'], r.get('train_step_ms') and round(r['train_step_ms']['value'],1), r['parity'] and r['parity']['net_within_1e-4'])"; done
grep hv_fwd_tiles $O/full_path_kernel_stats_default_streams.csv | cut -d, -f2-4 | tail -1
