"""Per-layer gradients at the training size, leaning on nothing from the path under test.

config 3 trains on batches of 3 scans (config/config.yaml:15): 3 x 80 000 = 240 000 rows at tensor stride 1.  The
end-to-end gradient test (tests/test_production_size_gpu.py) hands the HIP forward's ReLU masks to the oracle; here a
sample of layers of MinkUNet34C (utils/minkunet.py:53-120, utils/resnet.py:118-154) is checked one at a time on the
coordinate sets of such a batch: the HIP layer's own input x and upstream gradient dy (random, seeded) go to the
oracle's single-layer op in DOUBLE precision with torch autograd, and dX / dW / db / dgamma / dbeta must agree within
1e-4 of the largest gradient element.  No ReLU mask changes hands: convolutions have none, the BatchNorm + residual +
ReLU case zeroes dy wherever the ORACLE's pre-activation is within 1e-5 of zero, so an element whose sign the two
precisions could disagree on carries no gradient on either side.

The kernel maps are the HIP coordinate manager's (their exactness against the oracle's maps is asserted at <= 8k, 40k
and 300k rows in tests/test_sparse_gpu.py and tests/test_production_size_gpu.py); the arithmetic on them is what is
compared here.  Seconds of CPU per layer; always on."""
import zlib

import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def batch_cm(cuda):
    """coordinate manager of a 3 x 80k batch (train_joint.py:82 batched_coordinates)"""
    n = 80000
    scenes = [make_scene(40 + b, n_points=n) for b in range(3)]
    coords = np.concatenate([np.concatenate([np.full((n, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
    c = torch.from_numpy(coords).int().to(cuda)
    x = ME.SparseTensor(torch.zeros((len(coords), 1), device=cuda), c, device=cuda)
    cm = x.coordinate_manager
    cm.ensure_levels()
    return cm


def rel(g, ref):
    ref = ref.detach().numpy() if torch.is_tensor(ref) else ref
    return float(np.abs(g.detach().double().cpu().numpy() - ref).max() / max(1e-30, np.abs(ref).max()))


# (name in the network, kernel, stride, transposed, Cin, Cout, input tensor stride, bias)
CONVS = [
    ("conv0p1s1", 5, 1, False, 3, 32, 1, False),            # the 125-offset stem
    ("conv1p1s2", 2, 2, False, 32, 32, 1, False),           # strided: 240k -> level-2 rows
    ("block1.0.conv1", 3, 1, False, 32, 32, 2, False),      # mask-sorted groups (>= 16384 rows)
    ("block2.0.conv1", 3, 1, False, 32, 64, 4, False),
    ("block3.0.conv1", 3, 1, False, 64, 128, 8, False),     # split-K sizing
    ("block4.0.conv1", 3, 1, False, 128, 256, 16, False),
    ("block5.0.conv1", 3, 1, False, 384, 256, 8, False),    # the widest input (decoder concat)
    ("convtr7p2s2", 2, 2, True, 96, 96, 2, False),          # transposed: level-2 rows -> 240k
    ("block8.0.conv1", 3, 1, False, 128, 96, 1, False),     # the largest layer of the step
    ("final", 1, 1, False, 96, 64, 1, True),                # 1x1 + bias (utils/minkunet.py:114-119)
]


@pytest.mark.parametrize("name,k,stride,transposed,cin,cout,ts,bias", CONVS, ids=[c[0] for c in CONVS])
def test_conv_layer_gradients_at_three_80k_scenes(cuda, built_lib, batch_cm, name, k, stride, transposed, cin, cout, ts, bias):
    cm = batch_cm
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
    cls = ME.MinkowskiConvolutionTranspose if transposed else ME.MinkowskiConvolution
    torch.manual_seed(3)
    layer = cls(cin, cout, kernel_size=k, stride=stride, bias=bias, dimension=3).to(cuda).train()
    n_in = cm.num_rows(ts)
    x = torch.randn((n_in, cin), generator=g)
    xs = ME.SparseTensor(x.to(cuda).requires_grad_(True), coordinate_manager=cm, tensor_stride=ts)
    y = layer(xs)
    dy = torch.randn(tuple(y.F.shape), generator=g)
    (y.F * dy.to(cuda)).sum().backward()
    # the oracle's single-layer op in fp64 on the same map (a transposed k2s2 conv is a gather conv on the up map:
    # out[v] = W_oct(v)^T x[parent(v)])
    nbr, _ = layer._map(xs)
    nbr = nbr.cpu().numpy().astype(np.int64) if nbr is not None else np.arange(n_in, dtype=np.int64)[:, None]
    assert nbr.shape[0] == y.F.shape[0]
    xo = x.double().requires_grad_(True)
    wo = layer.kernel.detach().cpu().double().requires_grad_(True)
    bo = layer.bias.detach().cpu().double().requires_grad_(True) if bias else None
    yo = so.conv(xo, wo, nbr, bo)
    (yo * dy.double()).sum().backward()
    errs = {"y": rel(y.F, yo), "dx": rel(xs.F.grad, xo.grad), "dw": rel(layer.kernel.grad, wo.grad)}
    if bias:
        errs["db"] = rel(layer.bias.grad, bo.grad)
    pairs = int((nbr >= 0).sum())
    print("%s: rows %d -> %d, %d pairs, errors (max |d| / max |ref|): %s" % (
        name, n_in, nbr.shape[0], pairs, {k_: "%.1e" % v for k_, v in errs.items()}))
    assert all(v < TOL for v in errs.values()), errs


# (name, channels, tensor stride, residual, relu)
NORMS = [
    ("bn0", 32, 1, False, False),                 # 240k rows x 32
    ("block8.1.norm2", 96, 1, True, True),        # BasicBlock tail: relu(bn(x) + residual), the largest one
    ("block3.0.norm1", 128, 8, False, True),
]


@pytest.mark.parametrize("name,c,ts,residual,relu", NORMS, ids=[n[0] for n in NORMS])
def test_batchnorm_layer_gradients_at_three_80k_scenes(cuda, built_lib, batch_cm, name, c, ts, residual, relu):
    cm = batch_cm
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7fffffff)
    n = cm.num_rows(ts)
    x = torch.randn((n, c), generator=g) * (0.5 + torch.rand((1, c), generator=g)) + torch.randn((1, c), generator=g)
    res = torch.randn((n, c), generator=g) if residual else None
    gamma, beta = 0.5 + torch.rand(c, generator=g), 0.3 * torch.randn(c, generator=g)
    dy = torch.randn((n, c), generator=g)
    # the oracle first (fp64): batch-statistics BatchNorm1d (+ residual, ReLU) with autograd
    xo, go, bo = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ro = res.double().requires_grad_(True) if residual else None
    pre = torch.nn.functional.batch_norm(xo, None, None, go, bo, training=True, eps=1e-5)
    if residual:
        pre = pre + ro
    if relu:
        # no gradient where the sign of the pre-activation is not decided at fp32 accuracy (decided by the ORACLE's values)
        dy = dy * (pre.detach().abs() > 1e-5).float()
    yo = torch.relu(pre) if relu else pre
    (yo * dy.double()).sum().backward()
    bn = ME.MinkowskiBatchNorm(c).to(cuda).train()
    with torch.no_grad():
        bn.bn.weight.copy_(gamma)
        bn.bn.bias.copy_(beta)
    xs = ME.SparseTensor(x.to(cuda).requires_grad_(True), coordinate_manager=cm, tensor_stride=ts)
    rd = res.to(cuda).requires_grad_(True) if residual else None
    y = bn.forward_fused(xs, residual=rd, relu=relu)
    (y.F * dy.to(cuda)).sum().backward()
    errs = {"y": rel(y.F, yo), "dx": rel(xs.F.grad, xo.grad), "dgamma": rel(bn.bn.weight.grad, go.grad),
            "dbeta": rel(bn.bn.bias.grad, bo.grad)}
    if residual:
        errs["dres"] = rel(rd.grad, ro.grad)
    # running statistics as nn.BatchNorm1d updates them (momentum 0.1, unbiased variance)
    xm, xv = x.double().mean(0), x.double().var(0, unbiased=True)
    errs["running_mean"] = rel(bn.bn.running_mean, 0.1 * xm)
    errs["running_var"] = rel(bn.bn.running_var, 0.9 + 0.1 * xv)
    print("%s: %d rows x %d, errors: %s" % (name, n, c, {k_: "%.1e" % v for k_, v in errs.items()}))
    assert all(v < TOL for v in errs.values()), errs
