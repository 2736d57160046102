"""One fine-level 3x3x3 conv (96 -> 96, hl operands) on neighbour windows (conv_win), repeated - for rocprofv3 --pmc passes
and quick timing.  argv: repeats [rows = 80000] [ts = 1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
ts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cin = int(os.environ.get('MICRO_CIN', '96'))
cout = int(os.environ.get('MICRO_COUT', '96'))
sc = make_scene(3, n)
c4 = torch.cat([torch.zeros((n, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
nbr = cm.kernel_map(3, ts)
N = nbr.shape[0]
x = ME.to_hl(torch.randn(N, cin, device=dev))
w = torch.randn(27, cin, cout, device=dev) * 0.02
win = cm.windows(ts)
ME.set_option('win', 1)          # (off by default)
for _ in range(3):
    y = ME.conv_forward(x, w, nbr, N, relu=True, pieces=2, in_hl=True, out_hl=True, win=win)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = ME.conv_forward(x, w, nbr, N, relu=True, pieces=2, in_hl=True, out_hl=True, win=win)
e1.record()
torch.cuda.synchronize()
print("rows %d ts %d %d->%d: %.1f us per launch" % (N, ts, cin, cout, e0.elapsed_time(e1) / reps * 1e3))
