"""Shader-clock ticks per phase of conv_rows (CV_CONV_PROF=1 twin kernel) for the conv shapes of one 80k scene."""
import os, sys
os.environ['CV_CONV_PROF'] = '1'
os.environ['CV_NET_PROGRAM'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
sc = make_scene(3, 80000)
c4 = torch.cat([torch.zeros((80000, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
cm = ME.CoordinateManager(c4).fused_plan()[0]
for ts, cin, cout in ((1, 96, 96), (2, 96, 96), (4, 128, 128), (8, 256, 256), (16, 256, 256)):
    n = cm.num_rows(ts)
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.02
    nbr = cm.kernel_map(3, ts)
    for _ in range(2):
        if n >= 16384:
            ME.conv_forward_masked(x, w, nbr, cm.mask_perms(3, ts, 4), n, relu=True)
        else:
            ME.conv_forward(x, w, nbr, n, relu=True)
    torch.cuda.synchronize()
