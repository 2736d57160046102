"""large-row checks of the training kernels one layer at a time: BatchNorm (train) forward/backward vs torch, and a 3x3x3
conv's dX / dW vs the oracle's autograd at a masked-group size"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import me as ME
from oracle import sparse_oracle as so
from canonicalvoting_amd.synth import make_scene
dev = torch.device("cuda:0")
torch.manual_seed(0)
for n, c in [(2000, 96), (30000, 128), (240000, 96), (240000, 32)]:
    x = torch.randn(n, c, device=dev) * 2 + 0.5
    res = torch.randn(n, c, device=dev)
    gy = torch.randn(n, c, device=dev)
    bn = ME.MinkowskiBatchNorm(c).to(dev).train()
    with torch.no_grad():
        bn.bn.weight.uniform_(0.5, 1.5); bn.bn.bias.uniform_(-0.5, 0.5)
    xa = x.clone().requires_grad_(True); ra = res.clone().requires_grad_(True)
    st = ME.SparseTensor.__new__(ME.SparseTensor); st.F = xa; st.coordinate_manager = None; st.tensor_stride = 1
    st._like = lambda F, ts=None: F
    y = bn.forward_fused(st, residual=ra, relu=True)
    (y * gy).sum().backward()
    g_w, g_b = bn.bn.weight.grad.clone(), bn.bn.bias.grad.clone()
    xb = x.clone().double().requires_grad_(True); rb = res.clone().double().requires_grad_(True)
    w = bn.bn.weight.detach().double().requires_grad_(True); b = bn.bn.bias.detach().double().requires_grad_(True)
    yb = torch.relu(torch.nn.functional.batch_norm(xb, None, None, w, b, True, 0.1, bn.bn.eps) + rb)
    (yb * gy.double()).sum().backward()
    rel = lambda a, bb: float((a.double() - bb).abs().max() / bb.abs().max())
    print("BN train n=%6d c=%3d: y %.2e  dx %.2e  dres %.2e  dgamma %.2e  dbeta %.2e" % (n, c, rel(y, yb), rel(xa.grad, xb.grad), rel(ra.grad, rb.grad), rel(g_w, w.grad), rel(g_b, b.grad)))
# conv at a masked-group size: 3 x 30k rows
scenes = [make_scene(50 + b, n_points=30000) for b in range(3)]
coords = np.concatenate([np.concatenate([np.full((30000, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
cm = ME.CoordinateManager(torch.from_numpy(coords).to(dev, torch.int32))
ocm = so.CoordinateManager(coords)
rng = np.random.default_rng(0)
for cin, cout in [(32, 32), (96, 96)]:
    nbr, onbr = cm.kernel_map(3, 1), ocm.map(3, 1)
    n = len(coords)
    x = rng.normal(0, 1, (n, cin)).astype(np.float32); w = (rng.normal(0, 1, (27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    gy = rng.normal(0, 1, (n, cout)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True); wd = torch.from_numpy(w).to(dev).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, None, nbr, n)
    (y * torch.from_numpy(gy).to(dev)).sum().backward()
    xo = torch.from_numpy(x).requires_grad_(True); wo = torch.from_numpy(w).requires_grad_(True)
    yo = so.conv(xo, wo, onbr, None)
    (yo * torch.from_numpy(gy)).sum().backward()
    rel = lambda a, bb: float((a.detach().cpu() - bb.detach()).abs().max() / bb.detach().abs().max())
    print("conv k3 n=%d %d->%d: y %.2e  dX %.2e  dW %.2e" % (n, cin, cout, rel(y, yo), rel(xd.grad, xo.grad), rel(wd.grad, wo.grad)))
