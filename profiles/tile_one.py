"""One tile-kernel conv shape repeated (for rocprofv3 --pmc passes): python profiles/tile_one.py ts cin cout [flavour] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
ts, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flavour = int(sys.argv[4]) if len(sys.argv) > 4 else 4
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
sc = make_scene(3, 80000)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
n = cm.num_rows(ts)
x = torch.randn(n, cin, device=dev)
w = torch.randn(27, cin, cout, device=dev) * 0.02
nbr = cm.kernel_map(3, ts)
for _ in range(reps):
    y = ME.conv_forward(x, w, nbr, n, relu=True, flavour=flavour)
torch.cuda.synchronize()
print(float(y.abs().mean()))
