"""How far is an fp32 evaluation of one train_joint.py step from the exact result?  The oracle in fp32 against the oracle in
fp64 (ReLU masks of values within rounding of zero flip; a flipped mask changes a gradient element by its whole value),
then the same with the fp32 run's masks forced on the fp64 run: the reason
tests/test_production_size_gpu.py::test_config3_training_gradients_at_three_20k_scenes compares on one set of masks."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd.synth import make_scene
from canonicalvoting_amd import train
from oracle import sparse_oracle as so
N = int(sys.argv[1]); B = 3
scenes = [make_scene(60 + b, n_points=N) for b in range(B)]
coords = np.concatenate([np.concatenate([np.full((N, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
feats = np.concatenate([s.feats for s in scenes]).astype(np.float32) * 2 - 1
xyz, scale, cls = [np.concatenate([getattr(s, k) for s in scenes]) for k in ("xyz_labels", "scale_labels", "class_labels")]
sd = so.make_state_dict(3, 64, seed=1)
pn = [k for k in sd if k.split('.')[-1] in ('kernel', 'weight', 'bias') and sd[k].dtype.is_floating_point]
def run(dt, perturb=0.0):
    s = {k: (v.clone().to(dt).requires_grad_(True) if k in pn else v.clone()) for k, v in sd.items()}
    f = feats.astype(np.float64 if dt == torch.float64 else np.float32)
    y = so.minkunet34c_forward(s, coords, f, training=True, dtype=dt)
    l = train.joint_loss(y, torch.from_numpy(xyz).to(dt), torch.from_numpy(scale).to(dt), torch.from_numpy(cls))[0]
    l.backward()
    return float(l), {k: s[k].grad.double() for k in pn}
l64, g64 = run(torch.float64)
l32, g32 = run(torch.float32)
errs = sorted(((float((g32[k] - g64[k]).abs().max() / max(1e-12, float(g64[k].abs().max()))), k) for k in pn), reverse=True)
print("N", N, "loss", l64, l32)
print("fp32 oracle vs fp64 oracle, max|d|/max|g|: top", [(k, "%.2e" % e) for e, k in errs[:6]], "median %.2e" % errs[len(errs)//2][0])
# same activation pattern: the fp32 run's ReLU masks forced on the fp64 run
so.relu_trace = []
l32b, g32b = run(torch.float32)
masks = so.relu_trace; so.relu_trace = None
flips = 0
so.relu_masks = iter(masks)
l64m, g64m = run(torch.float64)
so.relu_masks = None
errs = sorted(((float((g32b[k] - g64m[k]).abs().max() / max(1e-12, float(g64m[k].abs().max()))), k) for k in pn), reverse=True)
print("same ReLU masks (%d ReLUs): fp32 oracle vs fp64 oracle: top" % len(masks), [(k, "%.2e" % e) for e, k in errs[:4]], "median %.2e" % errs[len(errs)//2][0])
