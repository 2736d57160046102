# BASELINE configs 1 and 5 shaped scenes through the default path (8k points; 300k points in the 9 x 3 x 9 m room)
cd $GRAFT_REPO_ROOT
python bench.py --points 8000 --steps 60 --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('8k six in flight', round(d['value'],1), d['stage_ms'], round(d['roofline']['frac'],3))"
python bench.py --large --points 300000 --steps 30 --warmup 4 --cpu-scenes 0 --streams 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('300k three in flight', round(d['value'],1), d['stage_ms'], round(d['roofline']['frac'],3), round(d['roofline_conv']['frac'],3), d['config']['grid'])"
python bench.py --large --points 300000 --steps 20 --warmup 4 --cpu-scenes 0 --streams 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('300k one in flight', round(d['value'],1), d['stage_ms'], round(d['roofline']['frac'],3))"
