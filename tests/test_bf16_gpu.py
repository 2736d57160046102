"""Opt-in bf16 compute mode of the sparse convolutions (BASELINE configs 3-4 name bf16 for training; the reference
itself is fp32, train_joint.py:218): operands rounded to bf16 (RNE), ONE bf16 x bf16 product on
v_mfma_f32_32x32x16_bf16, fp32 accumulation and storage.  It is NOT a parity path, so the tests pin what it IS:
exactly the fp32 path applied to bf16-rounded operands (forward, input gradient, weight gradient), a network output
at bf16 distance from the fp32 one, and a training step that still learns."""
import numpy as np
import pytest
import torch

from canonicalvoting_amd import me as ME
from canonicalvoting_amd.minkunet import MinkUNet34C
from tests.test_sparse_gpu import scene_coords

pytestmark = pytest.mark.gpu


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("cin,cout", [(32, 32), (96, 96), (128, 256)])
def test_bf16_conv_is_the_fp32_conv_of_rounded_operands(cuda, built_lib, cin, cout):
    coords, _ = scene_coords(51, 3000, small=False)
    n = len(coords)
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.randn(n, cin, generator=g) * torch.exp(torch.randn(n, cin, generator=g))).to(cuda)
    w = (torch.randn(27, cin, cout, generator=g) / 30).to(cuda)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr = cm.kernel_map(3, 1)
    assert ME.COMPUTE_DTYPE == "fp32"
    for kw in (dict(flavour=1), dict()):                       # one launch / offsets split over workgroups
        got = ME.conv_forward(x, w, nbr, n, pieces=1, **kw)
        want = ME.conv_forward(bf16_round(x), bf16_round(w), nbr, n, pieces=3, **kw)
        exact = ME.conv_forward(x, w, nbr, n, pieces=3, **kw)
        scale = float(want.abs().max())
        # same exact products, fp32 accumulation in both: only the summation order differs
        assert float((got - want).abs().max()) < 2e-5 * scale, kw
        # and it really is a bf16 computation: visibly away from the fp32 result, by about 2^-8 relative
        d = float((got - exact).abs().max()) / scale
        assert 1e-5 < d < 3e-2, d
    # mask-grouped launch (what the fine levels of the network use)
    perms = cm.mask_perms(3, 1, 4)
    got = ME.conv_forward_masked(x, w, nbr, perms, n, pieces=1)
    want = ME.conv_forward_masked(bf16_round(x), bf16_round(w), nbr, perms, n, pieces=3)
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("cin,cout", [(32, 64), (96, 96)])
def test_bf16_backward_is_the_fp32_backward_of_rounded_operands(cuda, built_lib, cin, cout):
    coords, _ = scene_coords(52, 1500)
    n = len(coords)
    g = torch.Generator().manual_seed(7 * cin + cout)
    x = torch.randn(n, cin, generator=g).to(cuda)
    w = (torch.randn(27, cin, cout, generator=g) / 30).to(cuda)
    gy = torch.randn(n, cout, generator=g).to(cuda)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr = cm.kernel_map(3, 1)

    def grads(xv, wv, gv):
        xd, wd = xv.clone().requires_grad_(True), wv.clone().requires_grad_(True)
        y = ME._ConvFn.apply(xd, wd, None, nbr, n)
        (y * gv).sum().backward()
        return y.detach(), xd.grad, wd.grad

    prev = ME.set_compute_dtype("bf16")
    try:
        y1, dx1, dw1 = grads(x, w, gy)
    finally:
        ME.set_compute_dtype(prev)
    y3, dx3, dw3 = grads(bf16_round(x), bf16_round(w), bf16_round(gy))
    for name, a, b in (("y", y1, y3), ("dX", dx1, dx3), ("dW", dw1, dw3)):
        assert float((a - b).abs().max()) < 3e-5 * float(b.abs().max()), name
    _, dx, dw = grads(x, w, gy)                                 # fp32 path on the unrounded operands
    for name, a, b in (("dX", dx1, dx), ("dW", dw1, dw)):
        d = float((a - b).abs().max()) / float(b.abs().max())
        assert 1e-6 < d < 3e-2, (name, d)


def test_bf16_network_forward_and_training_step(cuda, built_lib):
    from canonicalvoting_amd import train
    from canonicalvoting_amd.synth import make_scene
    coords, feats = scene_coords(53, 6000, small=False)
    torch.manual_seed(1)
    model = MinkUNet34C(3, 64).cuda().eval()
    x = lambda: ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        y32 = model(x()).F.clone()
        prev = ME.set_compute_dtype("bf16")
        try:
            y16 = model(x()).F.clone()
            y16_again = model(x()).F.clone()
        finally:
            ME.set_compute_dtype(prev)
        y32_again = model(x()).F.clone()
    assert torch.equal(y32, y32_again)                          # switching modes does not leak into the fp32 programs
    assert torch.equal(y16, y16_again)
    d = float((y16 - y32).abs().max()) / float(y32.abs().max())
    assert 1e-5 < d < 0.1, d                                    # bf16 distance over 60 stacked convolutions
    # training: the bf16 step learns, and its gradients point the way the fp32 ones do
    scenes = [make_scene(30 + b, n_points=900, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
              for b in range(3)]
    c4 = torch.cat([torch.cat([torch.full((900, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                    for b, s in enumerate(scenes)]).to(cuda)
    f = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(cuda) * 2 - 1
    xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(cuda)
    scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(cuda)
    cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(cuda)

    def grads_of(dtype):
        torch.manual_seed(0)
        m = MinkUNet34C(3, 64).cuda().train()
        prev = ME.set_compute_dtype(dtype)
        try:
            out = m(ME.SparseTensor(f, c4, device="cuda")).F
            loss, _ = train.joint_loss(out, xyz, scale, cls)
            loss.backward()
        finally:
            ME.set_compute_dtype(prev)
        return float(loss.detach()), torch.cat([p.grad.flatten() for p in m.parameters()])

    l32, g32 = grads_of("fp32")
    l16, g16 = grads_of("bf16")
    assert abs(l16 - l32) < 0.05 * abs(l32)
    cos = float(torch.dot(g16, g32) / (g16.norm() * g32.norm()))
    # 0.91 measured on MI355X for this random-init 60-layer network on 3 x 900 points (batch-statistics BatchNorm on
    # a few hundred coarse rows amplifies the 2^-8 operand rounding); an unrelated direction would be ~0
    assert cos > 0.8, cos
    torch.manual_seed(0)
    m = MinkUNet34C(3, 64).cuda().train()
    opt = train.make_optimizer(m, lr=1e-3)
    prev = ME.set_compute_dtype("bf16")
    try:
        hist = [float(train.train_step(m, opt, c4, f, xyz, scale, cls)[0]) for _ in range(6)]
    finally:
        ME.set_compute_dtype(prev)
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
