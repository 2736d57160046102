"""Weight-gradient kernel on representative MinkUNet34C layers of a 3 x 80k-point training batch: time per
launch and achieved TFLOP/s over the existing (input, output) pairs.  python profiles/wgrad_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
B, N = 3, 80000
scenes = [make_scene(b, n_points=N) for b in range(B)]
c4 = torch.cat([torch.cat([torch.full((N, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                for b, s in enumerate(scenes)]).to(dev)
cm = ME.CoordinateManager(c4)
layers = [(1, 96, 96), (1, 128, 96), (2, 32, 32), (2, 96, 96), (4, 64, 64), (8, 128, 128), (8, 384, 256), (16, 256, 256)]
tot = 0.0
for ts, cin, cout in layers:
    n = cm.num_rows(ts)
    nbr = cm.kernel_map(3, ts)
    pairs = int((nbr >= 0).sum())
    x = torch.randn(n, cin, device=dev)
    dy = torch.randn(n, cout, device=dev)
    for _ in range(3):
        dw = ME.conv_wgrad(x, dy, nbr, 27)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dw = ME.conv_wgrad(x, dy, nbr, 27)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    print("ts%-2d %3d->%3d n=%7d pairs/row=%5.2f  %7.3f ms  %6.2f TF/s" % (ts, cin, cout, n, pairs / n, ms, 2.0 * pairs * cin * cout / ms / 1e9))
print("sum %.3f ms" % tot)
