import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from canonicalvoting_amd import hv_cuda
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions
dev = torch.device('cuda:0')
orig_empty = torch.empty
def nan_empty(*a, **k):
    t = orig_empty(*a, **k)
    if t.is_floating_point() and t.is_cuda and t.dim() >= 3:
        t.fill_(float('nan'))
    return t
hv_cuda.torch.empty = nan_empty
for (n, res, kw) in [(1750, 0.06, dict(room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5)), (80000, 0.03, {}), (8000, 0.03, {})]:
    sc = make_scene(1, n_points=n, res=res, **kw)
    xyz, scale, prob, cls = synth_predictions(sc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    hv = HoughVoting(res, 120)
    with torch.no_grad():
        g = hv(t(sc.points), t(xyz), t(scale), t(prob))
    torch.cuda.synchronize()
    print(n, [tuple(x.shape) for x in g], 'NaN cells:', [int(torch.isnan(x).sum()) for x in g])
    if int(torch.isnan(g[0]).sum()):
        idx = torch.isnan(g[0]).nonzero()
        print('  first unwritten cells', idx[:10].tolist(), 'x range', int(idx[:,0].min()), int(idx[:,0].max()), 'y', int(idx[:,1].min()), int(idx[:,1].max()), 'z', int(idx[:,2].min()), int(idx[:,2].max()))
