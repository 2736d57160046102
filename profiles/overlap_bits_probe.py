"""Which parameter gradients differ between two training runs: one stream twice (control) and one stream vs the
side-stream weight gradient (ME.BACKWARD_OVERLAP)."""
import sys

import numpy as np
import torch
from oracle import sparse_oracle as so
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.minkunet import MinkUNet34C
from tests.test_sparse_gpu import scene_coords

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2500          # points per scene (80000: the production size)
coords, feats = scene_coords(17, N, batch=3, small=N <= 5000)
n = len(coords)
sd = so.make_state_dict(3, 64, seed=9)
rng = np.random.default_rng(1)
tgt = torch.from_numpy(rng.normal(0, 1, (n, 54)).astype(np.float32)).cuda()
labels = torch.from_numpy(rng.integers(0, 10, n)).cuda()


def run(overlap, steps=3):
    ME.BACKWARD_OVERLAP = overlap
    model = MinkUNet34C(3, 64)
    model.load_state_dict(sd)
    model = model.cuda().train()
    for _ in range(steps):
        model.zero_grad(set_to_none=True)
        x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
        out = model(x).F
        loss = ((out[:, :54] - tgt) ** 2).mean() + torch.nn.functional.cross_entropy(out[:, 54:], labels)
        loss.backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in model.named_parameters()}


a, b, c, d = run(False), run(False), run(True), run(True)
for name, (p, q) in {"one stream vs one stream": (a, b), "one stream vs overlap": (a, c), "overlap vs overlap": (c, d)}.items():
    bad = [(k, float((p[k] - q[k]).abs().max() / p[k].abs().max().clamp_min(1e-30))) for k in p if not torch.equal(p[k], q[k])]
    print("3 x %d points," % N, name, ":", len(bad), "of", len(p), "differ", bad[:8])
