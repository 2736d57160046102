"""Pair-compacted tile kernel (flavour 4) against the output-stationary kernel (flavour 1 + offset splits / mask
groups) on the 3x3x3 conv shapes of the five levels of one 80k-point scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
sc = make_scene(3, 80000)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print('%4s %6s %4s %4s %9s %9s' % ('ts', 'rows', 'cin', 'cout', 'tile us', 'rows us'))
for ts, cin, cout in ((1, 96, 96), (1, 128, 96), (2, 32, 32), (2, 96, 96), (4, 64, 64), (4, 128, 128), (4, 192, 128),
                      (8, 128, 128), (8, 256, 256), (8, 384, 256), (16, 256, 256)):
    n = cm.num_rows(ts)
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.02
    nbr = cm.kernel_map(3, ts)
    t_tile = bench(lambda: ME.conv_forward(x, w, nbr, n, relu=True, flavour=4))
    if n >= 16384:
        perms = cm.mask_perms(3, ts, 4)
        t_rows = bench(lambda: ME.conv_forward_masked(x, w, nbr, perms, n, relu=True))
    else:
        t_rows = bench(lambda: ME.conv_forward(x, w, nbr, n, relu=True, flavour=0))
    print('%4d %6d %4d %4d %9.1f %9.1f' % (ts, n, cin, cout, t_tile, t_rows))
