"""Side measurement (not the headline bench): train_joint.py-shaped steps (fwd + bwd + Adam) on synthetic
batches.  python profiles/train_bench.py [--points 80000] [--batch 3] [--steps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import train
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=80000)
ap.add_argument("--batch", type=int, default=3)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda")
scenes = [make_scene(b, n_points=a.points) for b in range(a.batch)]
coords = torch.cat([torch.cat([torch.full((a.points, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                    for b, s in enumerate(scenes)]).to(dev)
feats = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(dev) * 2 - 1
xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(dev)
scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(dev)
cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(dev)
torch.manual_seed(0)
model = MinkUNet34C(3, 64).cuda().train()
opt = train.make_optimizer(model)
for _ in range(2):
    train.train_step(model, opt, coords, feats, xyz, scale, cls)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.steps):
    loss, _ = train.train_step(model, opt, coords, feats, xyz, scale, cls)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / a.steps
print("train step %.1f ms  (%d x %d points, fp32)  %.1f scenes/s  loss %.4f" % (dt * 1e3, a.batch, a.points, a.batch / dt, float(loss)))
