"""the vote is a sum over points: permuting the points must not change one bit of the grids (fixed-point accumulation).
A difference = an arrangement-dependent bug in the tile kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import pipeline
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = MinkUNet34C(3, 64).to(dev).eval()
hv = HoughVoting(0.06, 120)
for seed in range(4):
    sc = make_scene(seed, n_points=1500 + 250 * seed, res=0.06, room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5)
    c4 = torch.cat([torch.zeros((len(sc.coords), 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
    f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
    with torch.no_grad():
        y = model(ME.SparseTensor(f, c4, device=dev))
        xyz, scale, prob, cls = pipeline.head_joint(y.F)
        pts = (c4[:, 1:] * 0.06).float().contiguous()
        ref = hv(pts, xyz, scale, prob)
        bad = 0
        g = torch.Generator(device="cpu").manual_seed(seed)
        for k in range(30):
            p = torch.randperm(len(pts), generator=g).to(dev)
            out = hv(pts[p].contiguous(), xyz[p].contiguous(), scale[p].contiguous(), prob[p].contiguous())
            if not torch.equal(out[0], ref[0]):
                d = (out[0] != ref[0])
                bad += 1
                if bad <= 2:
                    print("   perm %d: %d cells differ, sum diff %.6g, cells %s" % (k, int(d.sum()), float(out[0].double().sum() - ref[0].double().sum()), d.nonzero()[:6].tolist()))
    print("scene %d: %d of 30 permutations change the grid" % (seed, bad))
