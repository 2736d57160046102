"""cProfile of the host side of train_joint.py steps (3 x 80k rows): where the Python time of a step goes."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from canonicalvoting_amd import train
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
scenes = [make_scene(40 + b, n_points=n) for b in range(3)]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
coords = t(np.concatenate([np.concatenate([np.full((n, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])).int()
feats = t(np.concatenate([s.feats for s in scenes]).astype(np.float32)) * 2 - 1
xyz, scale, cls = [t(np.concatenate([getattr(s, k) for s in scenes])) for k in ("xyz_labels", "scale_labels", "class_labels")]
torch.manual_seed(0)
model = MinkUNet34C(3, 64).cuda().train()
opt = train.make_optimizer(model)
for _ in range(3):
    train.train_step(model, opt, coords, feats, xyz, scale, cls)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    train.train_step(model, opt, coords, feats, xyz, scale, cls)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(30)
