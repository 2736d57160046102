# one-barrier-per-unit conv kernel (conv_rows_wp) against conv_rows_x6 (CV_CONV_WP=0): parity tests, layer times, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/wp
mkdir -p $O
timeout 1500 python -m pytest tests/test_sparse_gpu.py tests/test_train_gpu.py tests/test_bf16_gpu.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python - > $O/bitwise.txt 2>&1 <<'PY'
import os, subprocess, sys
code = """
import torch, numpy as np
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene
sc = make_scene(5, n_points=80000, res=0.03)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).cuda()
f = (torch.from_numpy(sc.feats) * 2 - 1).cuda()
torch.manual_seed(0)
m = MinkUNet34C(3, 64).cuda().eval()
with torch.no_grad():
    y = m(ME.SparseTensor(f, c4, device='cuda')).F
torch.save(y.cpu(), '/tmp/y_%s.pt' % __import__('os').environ.get('CV_CONV_WP', '1'))
"""
for v in ("0", "1"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CV_CONV_WP=v), check=True)
import torch
a, b = torch.load("/tmp/y_0.pt"), torch.load("/tmp/y_1.pt")
print("network output bit-identical between conv_rows_x6 and conv_rows_wp:", bool(torch.equal(a, b)), float((a - b).abs().max()))
PY
cat $O/bitwise.txt | tail -2
for v in 0 1; do
  CV_CONV_WP=$v python profiles/layer_times.py 2>&1 | tail -66 > $O/layer_times_wp$v.txt; tail -1 $O/layer_times_wp$v.txt
  CV_CONV_WP=$v python bench.py --streams 1 --cpu-scenes 0 --steps 120 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WP=$v one in flight', round(d['value'],1), d['stage_ms'])"
  CV_CONV_WP=$v python bench.py --cpu-scenes 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WP=$v six in flight', round(d['value'],1))"
done
