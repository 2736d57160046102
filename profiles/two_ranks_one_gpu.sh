# the multi-rank launch path of bench.py exactly as the driver starts it, on a 1-GPU box: two ranks share GPU 0 and talk gloo
# (RCCL refuses two ranks on one device); checks rendezvous, barriers, the max-over-ranks time and the whole-job value
cd $GRAFT_REPO_ROOT
export CV_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 80 --warmup 6 2>gpurun_out/two_ranks.err | tail -1 > gpurun_out/two_ranks_eval.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --mode train --steps 6 --warmup 2 2>>gpurun_out/two_ranks.err | tail -1 > gpurun_out/two_ranks_train.json
python bench.py --steps 80 --warmup 6 --cpu-scenes 0 2>/dev/null | tail -1 > gpurun_out/one_rank_eval.json
for f in two_ranks_eval two_ranks_train one_rank_eval; do python -c "import json; d=json.loads(open('gpurun_out/$f.json').read()); print('$f', d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],3), d['config'].get('parallelism'))"; done
tail -3 gpurun_out/two_ranks.err
