"""single-layer dX / dW of the sparse conv at mid-level sizes (rows x channels of MinkUNet34C's levels 2-4 for a 3 x 20k
batch) against the oracle's autograd: where does the 1-4 % gradient error of the whole network come from?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import me as ME
from oracle import sparse_oracle as so
from canonicalvoting_amd.synth import make_scene
dev = torch.device("cuda:0")
scenes = [make_scene(60 + b, n_points=20000) for b in range(3)]
coords = np.concatenate([np.concatenate([np.full((20000, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
cm = ME.CoordinateManager(torch.from_numpy(coords).to(dev, torch.int32))
ocm = so.CoordinateManager(coords)
rng = np.random.default_rng(0)
rel = lambda a, bb: float((a.detach().cpu() - bb.detach()).abs().max() / bb.detach().abs().max())
print("rows per level", [cm.num_rows(1 << i) for i in range(5)])
for ts, cin, cout in [(2, 32, 32), (4, 64, 64), (4, 192, 128), (4, 128, 128), (8, 64, 128), (8, 128, 128), (8, 384, 256), (16, 256, 256), (2, 128, 96)]:
    nbr, onbr = cm.kernel_map(3, ts), ocm.map(3, ts)
    n = nbr.shape[0]
    x = rng.normal(0, 1, (n, cin)).astype(np.float32); w = (rng.normal(0, 1, (27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    gy = rng.normal(0, 1, (n, cout)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, None, nbr, n)
    (y * torch.from_numpy(gy).to(dev)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True); wo = torch.from_numpy(w).requires_grad_(True)
    yo = so.conv(xo, wo, onbr, None)
    (yo * torch.from_numpy(gy)).sum().backward()
    print("k3 ts%-2d n=%6d %3d->%3d: y %.2e  dX %.2e  dW %.2e" % (ts, n, cin, cout, rel(y, yo), rel(xd.grad, xo.grad), rel(wd.grad, wo.grad)))
for ts, cin, cout in [(1, 32, 32), (2, 32, 32), (4, 64, 64), (8, 128, 128)]:
    nbr, onbr = cm.kernel_map(2, ts, 2), ocm.map(2, ts, 2)
    n_in, n_out = cm.num_rows(ts), nbr.shape[0]
    x = rng.normal(0, 1, (n_in, cin)).astype(np.float32); w = (rng.normal(0, 1, (8, cin, cout)) / np.sqrt(cin * 8)).astype(np.float32)
    gy = rng.normal(0, 1, (n_out, cout)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, None, nbr, n_out)
    (y * torch.from_numpy(gy).to(dev)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True); wo = torch.from_numpy(w).requires_grad_(True)
    yo = so.conv(xo, wo, onbr, None)
    (yo * torch.from_numpy(gy)).sum().backward()
    print("down ts%-2d n_in=%6d n_out=%6d %3d->%3d: y %.2e  dX %.2e  dW %.2e" % (ts, n_in, n_out, cin, cout, rel(y, yo), rel(xd.grad, xo.grad), rel(wd.grad, wo.grad)))
    # transposed (up): coarse -> fine
    up = cm.up_map(2 * ts)
    x = rng.normal(0, 1, (n_out, cout)).astype(np.float32); w = (rng.normal(0, 1, (8, cout, cin)) / np.sqrt(cout * 8)).astype(np.float32)
    gy = rng.normal(0, 1, (n_in, cin)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, None, up, n_in)
    (y * torch.from_numpy(gy).to(dev)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True); wo = torch.from_numpy(w).requires_grad_(True)
    yo = so.conv_transpose_k2s2(xo, wo, onbr)
    (yo * torch.from_numpy(gy)).sum().backward()
    print("up   ts%-2d n_in=%6d n_out=%6d %3d->%3d: y %.2e  dX %.2e  dW %.2e" % (2 * ts, n_out, n_in, cout, cin, rel(y, yo), rel(xd.grad, xo.grad), rel(wd.grad, wo.grad)))
for ts, cin, cout in [(4, 192, 128), (8, 384, 256), (1, 96, 64)]:
    n = cm.num_rows(ts)
    x = rng.normal(0, 1, (n, cin)).astype(np.float32); w = (rng.normal(0, 1, (cin, cout)) / np.sqrt(cin)).astype(np.float32)
    gy = rng.normal(0, 1, (n, cout)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, None, None, n)
    (y * torch.from_numpy(gy).to(dev)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True); wo = torch.from_numpy(w).requires_grad_(True)
    yo = xo @ wo
    (yo * torch.from_numpy(gy)).sum().backward()
    print("k1 ts%-2d n=%6d %3d->%3d: y %.2e  dX %.2e  dW %.2e" % (ts, n, cin, cout, rel(y, yo), rel(xd.grad, xo.grad), rel(wd.grad, wo.grad)))
