"""Backward of the sparse engine (train_joint.py:283 `loss.backward()`): HIP input/weight/bias gradients of
every conv kind and of the whole MinkUNet34C vs torch autograd through the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.minkunet import MinkUNet34C
from tests.test_sparse_gpu import rel_err, scene_coords

pytestmark = pytest.mark.gpu


def run_pair(cuda, coords, cin, cout, kind, seed=0):
    rng = np.random.default_rng(seed)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    ocm = so.CoordinateManager(coords)
    if kind == "k3":
        nbr, onbr, K, n_in, n_out = cm.kernel_map(3, 1), ocm.map(3, 1), 27, len(coords), len(coords)
    elif kind == "k5":
        nbr, onbr, K, n_in, n_out = cm.kernel_map(5, 1), ocm.map(5, 1), 125, len(coords), len(coords)
    elif kind == "k1":
        nbr, onbr, K, n_in, n_out = None, ocm.map(1, 1), 1, len(coords), len(coords)
    elif kind == "down":
        nbr, onbr, K = cm.kernel_map(2, 1, 2), ocm.map(2, 1, 2), 8
        n_in, n_out = len(coords), cm.num_rows(2)
    else:   # up: coarse -> fine
        nbr, K = cm.up_map(2), 8
        onbr = ocm.map(2, 1, 2)
        n_in, n_out = cm.num_rows(2), len(coords)
    x = rng.normal(0, 1, (n_in, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (K, cin, cout)) / np.sqrt(cin * K)).astype(np.float32)
    bias = rng.normal(0, 1, (1, cout)).astype(np.float32)
    gy = rng.normal(0, 1, (n_out, cout)).astype(np.float32)
    # HIP
    xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
    wd = torch.from_numpy(w if K > 1 else w[0]).to(cuda).requires_grad_(True)
    bd = torch.from_numpy(bias).to(cuda).requires_grad_(True)
    y = ME._ConvFn.apply(xd, wd, bd, nbr, n_out)
    (y * torch.from_numpy(gy).to(cuda)).sum().backward()
    # oracle autograd
    xo = torch.from_numpy(x).requires_grad_(True)
    wo = torch.from_numpy(w).requires_grad_(True)
    bo = torch.from_numpy(bias).requires_grad_(True)
    if kind == "up":
        yo = so.conv_transpose_k2s2(xo, wo, onbr) + bo
    else:
        yo = so.conv(xo, wo, onbr, bo)
    (yo * torch.from_numpy(gy)).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), yo.detach().numpy()) < 1e-5
    assert rel_err(xd.grad.cpu().numpy(), xo.grad.numpy()) < 1e-5, kind + " dX"
    assert rel_err(wd.grad.cpu().numpy().reshape(w.shape), wo.grad.numpy()) < 2e-5, kind + " dW"
    assert rel_err(bd.grad.cpu().numpy(), bo.grad.numpy()) < 1e-5, kind + " dB"


@pytest.mark.parametrize("cin,cout,kind", [(32, 32, "k3"), (64, 96, "k3"), (128, 256, "k3"), (3, 32, "k5"),
                                           (96, 64, "k1"), (384, 256, "k1"), (32, 32, "down"), (128, 128, "down"),
                                           (256, 128, "up"), (96, 96, "up")])
def test_conv_backward_matches_oracle_autograd(cuda, built_lib, cin, cout, kind):
    coords, _ = scene_coords(11, 1200)
    run_pair(cuda, coords, cin, cout, kind, seed=cin + cout)


def test_minkunet_training_step_gradients_match_oracle(cuda, built_lib, monkeypatch):
    """train-mode forward (batch-statistics BN) + backward of a masked MSE/CE loss shaped like
    train_joint.py:253-283; every parameter gradient vs autograd through the CPU oracle.  Nothing here shares ReLU masks
    with the oracle, so the bar is set by which side of zero a few pre-activations of the 1400 rows land on, i.e. by the size
    of the forward's rounding: the test runs the forward on the bf16 triples (24 significant bits, ME.TRAIN_FWD_HL = 0), where
    every gradient is within 2e-3 of its maximum.  The default forward (fp16 pairs on the hl-format kernels, 22 bits) flips
    about four times as many ReLUs here (median parameter error 6e-3) - its products are pinned on SHARED masks instead:
    test_training_gradients_on_shared_masks_small below and, at 3 x 20k rows, tests/test_production_size_gpu.py (1e-4 bar,
    7e-6 measured)."""
    monkeypatch.setattr(ME, "TRAIN_FWD_HL", 0)
    bar = 2e-3
    coords, feats = scene_coords(13, 700, batch=2)
    n = len(coords)
    sd = so.make_state_dict(3, 64, seed=5)
    model = MinkUNet34C(3, 64)
    model.load_state_dict(sd)
    model = model.cuda().train()
    rng = np.random.default_rng(0)
    tgt = rng.normal(0, 1, (n, 54)).astype(np.float32)
    labels = rng.integers(0, 10, n)
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    out = model(x).F
    loss = ((out[:, :54] - torch.from_numpy(tgt).to(cuda)) ** 2).mean() + torch.nn.functional.cross_entropy(
        out[:, 54:], torch.from_numpy(labels).to(cuda))
    loss.backward()
    pnames = {k for k, _ in model.named_parameters()}
    sdo = {k: (v.clone().requires_grad_(True) if k in pnames else v.clone()) for k, v in sd.items()}
    yo = so.minkunet34c_forward(sdo, coords, feats, training=True)
    lo = ((yo[:, :54] - torch.from_numpy(tgt)) ** 2).mean() + torch.nn.functional.cross_entropy(
        yo[:, 54:], torch.from_numpy(labels))
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))
    worst, errs = 0.0, []
    for name, p in model.named_parameters():
        g, go = p.grad.cpu().numpy(), sdo[name].grad.numpy()
        err = np.abs(g - go).max() / max(1e-6, np.abs(go).max())
        worst = max(worst, err)
        errs.append(err)
        assert err < bar, (name, err)
    assert worst > 0 and np.median(errs) < 2e-3


def test_training_gradients_on_shared_masks_small(cuda, built_lib):
    """the default training forward (hl-format kernels) at 3 x 1500 rows - every level below the mask-sorting threshold, unsplit
    and split launches - all parameter gradients against the fp64 oracle on the forward's own ReLU masks, 1e-4 bar"""
    from tests.test_production_size_gpu import _training_gradients_on_shared_relu_masks
    assert ME.TRAIN_FWD_HL == 1
    _training_gradients_on_shared_relu_masks(cuda, 1500, 80)


def test_backward_overlap_changes_no_gradient_bit(cuda, built_lib, monkeypatch):
    """_ConvFn.backward runs a layer's weight gradient on a side stream next to its input gradient
    (ME.BACKWARD_OVERLAP: 1 joins the streams per layer, 2 at the end of the backward pass).  Same kernels, same
    inputs: every parameter gradient of a training step (three scenes in a batch, three steps so that freed blocks are
    reused across the two streams) equals the one-stream run bit for bit; a gradient accumulated over two passes
    (which mode 2 must not defer) does too."""
    coords, feats = scene_coords(17, 2500, batch=3)
    n = len(coords)
    sd = so.make_state_dict(3, 64, seed=9)
    rng = np.random.default_rng(1)
    tgt = torch.from_numpy(rng.normal(0, 1, (n, 54)).astype(np.float32)).to(cuda)
    labels = torch.from_numpy(rng.integers(0, 10, n)).to(cuda)
    grads = {}
    for overlap in (0, 1, 2):
        monkeypatch.setattr(ME, "BACKWARD_OVERLAP", overlap)
        model = MinkUNet34C(3, 64)
        model.load_state_dict(sd)
        model = model.cuda().train()
        for it in range(4):
            if it != 3:                        # the last pass accumulates into the third one's gradients
                model.zero_grad(set_to_none=True)
            x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
            out = model(x).F
            loss = ((out[:, :54] - tgt) ** 2).mean() + torch.nn.functional.cross_entropy(out[:, 54:], labels)
            loss.backward()
        torch.cuda.synchronize()
        grads[overlap] = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert len(ME._wgrad_streams) >= 1
    for mode in (1, 2):
        for k, g in grads[0].items():
            assert torch.isfinite(g).all(), k
            # (final.bias included: its column sums are added in a fixed order since cv_sp_col_sum_det_f32)
            assert torch.equal(g, grads[mode][k]), (mode, k)


def test_late_joined_weight_gradient_is_the_tensor_the_side_stream_wrote(cuda, built_lib, monkeypatch):
    """Mode 2 of the backward overlap leaves d_kernel to a side stream until the end of the pass.  That is only sound if
    the gradient accumulator KEEPS that tensor (no clone or add on the layer's stream): after the pass every late-joined
    layer's .grad must be the very storage the side stream wrote.  Fails if a torch change (or a layout the predicate
    misses) makes AccumulateGrad copy instead."""
    monkeypatch.setattr(ME, "BACKWARD_OVERLAP", 2)
    log = []
    monkeypatch.setattr(ME, "LATE_GRAD_LOG", log)
    coords, feats = scene_coords(23, 1500, batch=2)
    model = MinkUNet34C(3, 64)
    model.load_state_dict(so.make_state_dict(3, 64, seed=4))
    model = model.cuda().train()
    model.zero_grad(set_to_none=True)
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    model(x).F.square().mean().backward()
    torch.cuda.synchronize()
    assert len(log) >= 40, "the late join was not taken (%d layers)" % len(log)
    for kernel, ptr in log:
        assert kernel.grad is not None and kernel.grad.data_ptr() == ptr and kernel.grad.is_contiguous()
    # a second pass accumulates into existing gradients: nothing may be deferred then
    del log[:]
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    model(x).F.square().mean().backward()
    assert log == []


def test_column_sums_are_the_same_bits_on_every_run(cuda, built_lib):
    """ME.col_sum (bias gradient, cv_sp_col_sum_det_f32): chunk sums added in chunk order - equal bits over repeated
    runs and on another stream with other kernels in flight, and the fp64 column sums to fp32 accuracy; strided rows and
    a column count that is not a multiple of 32 included."""
    g = torch.Generator(device="cpu").manual_seed(5)
    for n, c, ld in ((240000, 64, 64), (70001, 37, 48), (5, 3, 3)):
        x = (torch.randn((n, ld), generator=g) * torch.logspace(-3, 3, ld)).to(cuda)[:, :c]
        want = x.double().sum(0)
        first = ME.col_sum(x)
        err = ((first.double() - want).abs() / x.double().abs().sum(0).clamp_min(1e-30)).max()
        assert float(err) < 2e-6, (n, c, float(err))
        side = torch.cuda.Stream()
        noise = torch.randn((4096, 4096), device=cuda)
        for _ in range(5):
            assert torch.equal(ME.col_sum(x), first)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                (noise @ noise).sum()
                again = ME.col_sum(x)
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(again, first)


def test_train_step_reduces_loss_and_matches_reference_loss(cuda, built_lib):
    """A few Adam steps on one synthetic batch (train_joint.py:246-288): the loss restated in
    canonicalvoting_amd.train equals a line-by-line torch restatement and goes down."""
    from canonicalvoting_amd import train
    from canonicalvoting_amd.synth import make_scene
    scenes = [make_scene(20 + b, n_points=900, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
              for b in range(3)]                                       # batch_size 3 (config.yaml:15)
    coords = torch.cat([torch.cat([torch.full((900, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                        for b, s in enumerate(scenes)]).to(cuda)
    feats = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(cuda) * 2 - 1
    xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(cuda)
    scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(cuda)
    cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(cuda)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    # loss parity on a fixed output
    out = torch.randn(len(cls), 64, device=cuda)
    loss, parts = train.joint_loss(out, xyz, scale, cls)
    lab = cls.long()
    idx = lab.clone(); idx[idx < 0] = 0; idx[idx == 9] = 0
    g = idx[:, None, None].expand(-1, -1, 3)
    oxyz = torch.gather(out[:, :27].reshape(-1, 9, 3), 1, g)[:, 0]
    oscale = torch.gather(out[:, 27:54].reshape(-1, 9, 3), 1, g)[:, 0]
    mask = (lab < 9) & (lab >= 0)
    ref = torch.mean((oscale[mask] - torch.log(scale[mask])) ** 2) + torch.mean((oxyz[mask] - xyz[mask]) ** 2) \
        + torch.nn.functional.cross_entropy(out[:, 54:], lab)
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    opt = train.make_optimizer(model, lr=1e-3)
    hist = [float(train.train_step(model, opt, coords, feats, xyz, scale, cls)[0]) for _ in range(6)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0]
    assert train.adjust_learning_rate(opt, 85) == pytest.approx(1e-4) and opt.param_groups[0]["lr"] == pytest.approx(1e-4)


def _small_batch(cuda, seed0=30, n=900):
    from canonicalvoting_amd.synth import make_scene
    scenes = [make_scene(seed0 + b, n_points=n, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4) for b in range(3)]
    coords = torch.cat([torch.cat([torch.full((n, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                        for b, s in enumerate(scenes)]).to(cuda)
    feats = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(cuda) * 2 - 1
    xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(cuda)
    scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(cuda)
    cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(cuda)
    return coords, feats, xyz, scale, cls


def test_training_forward_on_the_hl_kernels(cuda, built_lib, monkeypatch):
    """ME.TRAIN_FWD_HL (default): every BatchNorm + ReLU pass of the training forward writes its output a second time as fp16
    pairs (cv_sp_affine_hl_f32) and the convolution that reads it runs on the eval path's hl-format kernels.  The twin holds
    exactly the bits ME.to_hl gives, the network output agrees with the forward on the bf16 triples within 1e-5 of its largest
    value (a ReLU is continuous: a flipped pre-activation moves the OUTPUT by its own tiny value), the loss follows, and the
    saved tensors of autograd are the fp32 rows (the weight gradients of the two modes agree to the same level on a layer
    that no ReLU precedes... the stem's output feeds BatchNorm directly)."""
    coords, feats, xyz, scale, cls = _small_batch(cuda)
    from canonicalvoting_amd import train
    # the twin of one pass
    torch.manual_seed(1)
    bn = ME.MinkowskiBatchNorm(64).cuda().train()
    x = ME.SparseTensor(torch.randn(len(coords), 64, device=cuda) * 3, coords, device=cuda)
    res = torch.randn(len(coords), 64, device=cuda)
    monkeypatch.setattr(ME, "TRAIN_FWD_HL", 1)
    y = bn.forward_fused(x, residual=res, relu=True)
    assert y.F_hl is not None and torch.equal(y.F_hl.view(torch.int32), ME.to_hl(y.F.detach()).view(torch.int32))
    assert bn.forward_fused(x, relu=False).F_hl is None              # only what a convolution reads gets a twin
    z = ME.cat(y, y)
    assert torch.equal(ME.from_hl(z.F_hl), ME.from_hl(ME.to_hl(z.F.detach())))
    monkeypatch.setattr(ME, "TRAIN_FWD_HL", 0)
    y0 = bn.forward_fused(x, residual=res, relu=True)
    assert y0.F_hl is None and torch.equal(y0.F, y.F)
    # the network
    outs, grads = [], []
    for hl in (1, 0):
        monkeypatch.setattr(ME, "TRAIN_FWD_HL", hl)
        torch.manual_seed(0)
        model = MinkUNet34C(3, 64).cuda().train()
        out = model(ME.SparseTensor(feats, coords, device=cuda)).F
        loss, _ = train.joint_loss(out, xyz, scale, cls)
        loss.backward()
        outs.append((out.detach(), float(loss)))
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    (o1, l1), (o0, l0) = outs
    assert float((o1 - o0).abs().max()) < 1e-5 * max(1.0, float(o0.abs().max()))
    assert abs(l1 - l0) < 1e-5 * abs(l0)
    assert float(o0.abs().max()) > 1e-3
    # the last layer's gradients do not pass a ReLU on their way back: the two modes agree to product precision there
    for k in ("final.kernel", "final.bias"):
        assert float((grads[0][k] - grads[1][k]).abs().max()) < 2e-5 * float(grads[1][k].abs().max()), k


def test_input_gradients_on_the_hl_kernels(cuda, built_lib, monkeypatch):
    """ME.TRAIN_BWD_HL (default): inside train.train_step the BatchNorm backward writes dx a second time as fp16 pairs, scaled by
    the power of two that the layer's largest |dx| of the step BEFORE suggests, and the input gradient of the convolution below
    runs on the eval path's hl-format kernels (transposed map, W^T packed as fp16 pairs, the inverse factor read on the device).
    Same forward, same ReLU masks: every parameter gradient of the second step agrees with the run whose input gradients stay
    on the bf16 triples to product precision, the first step (no maximum yet) is bit-identical, and most convolutions took the
    new path."""
    from canonicalvoting_amd import train
    batch = _small_batch(cuda, seed0=50, n=1500)
    runs = []
    for bwd in (1, 0):
        monkeypatch.setattr(ME, "TRAIN_BWD_HL", bwd)
        torch.manual_seed(0)
        model = MinkUNet34C(3, 64).cuda().train()
        opt = train.make_optimizer(model, lr=0.0)                    # (the weights stay: both steps see the same network)
        ME.TRAIN_COUNTERS["hl_dgrad"] = 0
        train.train_step(model, opt, *batch)
        g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
        n1 = ME.TRAIN_COUNTERS["hl_dgrad"]
        train.train_step(model, opt, *batch)
        g2 = {k: p.grad.clone() for k, p in model.named_parameters()}
        runs.append((g1, g2, n1, ME.TRAIN_COUNTERS["hl_dgrad"] - n1))
    (a1, a2, a_n1, a_n2), (b1, b2, b_n1, b_n2) = runs
    assert a_n1 == 0 and a_n2 >= 55 and b_n1 == 0 and b_n2 == 0
    for k in a1:
        assert torch.equal(a1[k], b1[k]), k                           # no maximum yet: the same kernels
        assert torch.equal(b1[k], b2[k]), k                           # (and a step is reproducible bit for bit)
        d = float((a2[k] - b2[k]).abs().max())
        assert d < 2e-5 * max(1e-12, float(b2[k].abs().max())), (k, d)
    assert any(not torch.equal(a2[k], b2[k]) for k in a2)             # it IS another kernel


def test_weights_packed_once_per_step_are_the_same_bits(cuda, built_lib, monkeypatch):
    """ME.TRAIN_PREPACK (default): train.train_step packs the fp16-pair weights of all convolutions - forward layout and, for
    the input gradients, transposed - by ONE launch in front of the forward (cv_sp_pack_weights_h2_batch_f32) instead of one
    launch per layer and direction.  Same scales, same split: losses and every parameter gradient of two steps are bit-identical
    to the per-layer packing, and the second step takes > 100 packed tensors from the batch."""
    from canonicalvoting_amd import train
    batch = _small_batch(cuda, seed0=60, n=1500)
    runs = []
    for pre in (1, 0):
        monkeypatch.setattr(ME, "TRAIN_PREPACK", pre)
        torch.manual_seed(0)
        model = MinkUNet34C(3, 64).cuda().train()
        opt = train.make_optimizer(model, lr=1e-3)
        ME.TRAIN_COUNTERS["prepacked"] = 0
        out = []
        for _ in range(2):
            loss, _ = train.train_step(model, opt, *batch)
            out.append((float(loss), {k: p.grad.clone() for k, p in model.named_parameters()}))
        runs.append((out, ME.TRAIN_COUNTERS["prepacked"]))
    (a, n_a), (b, n_b) = runs
    assert n_b == 0 and n_a > 100, (n_a, n_b)
    for (la, ga), (lb, gb) in zip(a, b):
        assert la == lb
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


def test_a_forward_beyond_the_fp16_range_skips_its_update_on_the_device(cuda, built_lib, monkeypatch):
    """train.train_step with the fused Adam: no host wait in the step.  A BatchNorm gain of 1e7 puts activations beyond 65000:
    the hl twin raises the range flag, the optimizer gets it as found_inf and leaves every parameter (and its step count)
    alone; the call FLAG_LAG steps later notices, restores the BatchNorm running statistics and counters of before the first
    flagged step (the flagged forwards wrote inf / NaN statistics downstream of the overflow), counts the fallback,
    runs on the bf16 triples from then on and updates: every flagged batch is dropped as a whole, the redone one counts once.
    With an optimizer that cannot skip on the device (SGD) the step is redone on the triples before the optimizer sees a
    gradient, and the BatchNorm running statistics count the batch once."""
    from canonicalvoting_amd import train
    batch = _small_batch(cuda, seed0=40)
    monkeypatch.setattr(ME, "TRAIN_FWD_HL", 1)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    opt = train.make_optimizer(model, lr=1e-3)
    train.train_step(model, opt, *batch)                               # healthy
    assert getattr(model, "train_range_fallbacks", 0) == 0
    with torch.no_grad():
        model.bn0.bn.weight.fill_(1e7)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    torch.cuda.synchronize()
    stats_before = {k: b.detach().clone() for k, b in model.named_buffers()}
    nbt = int(model.bn0.bn.num_batches_tracked)
    for _ in range(train.FLAG_LAG):
        loss, _ = train.train_step(model, opt, *batch)
        torch.cuda.synchronize()
        assert all(torch.equal(before[k], p.detach()) for k, p in model.named_parameters())   # skipped on the device
        assert getattr(model, "train_range_fallbacks", 0) == 0                               # ... and nobody has looked yet
    # the flagged forwards DID poison the running statistics behind the overflow - what the snapshot is for
    assert not all(bool(torch.isfinite(b).all()) for b in model.buffers() if b.is_floating_point())
    # the snapshot guard: still the statistics of before the FIRST flagged step, although two flagged steps have run
    bufs, saved = train._bn_state(model)
    names = [k for k, _ in model.named_buffers()]
    assert len(names) == len(saved) and all(torch.equal(stats_before[k], s) for k, s in zip(names, saved))
    loss, _ = train.train_step(model, opt, *batch)
    torch.cuda.synchronize()
    assert model.train_range_fallbacks == 1 and np.isfinite(float(loss))
    assert any(not torch.equal(before[k], p.detach()) for k, p in model.named_parameters())   # the triples step went through
    assert int(model.bn0.bn.num_batches_tracked) == nbt + 1            # the flagged batches dropped, the redone one counted once
    assert all(int(m.bn.num_batches_tracked) == nbt + 1 for m in model.modules() if isinstance(m, ME.MinkowskiBatchNorm))
    assert all(bool(torch.isfinite(b).all()) for b in model.buffers())                        # no NaN statistics survive
    assert int(ME.range_flag(torch.device(cuda))[0]) == 0
    loss, _ = train.train_step(model, opt, *batch)                     # and the next step is an ordinary (triples) step
    assert model.train_range_fallbacks == 1 and np.isfinite(float(loss)) and int(model.bn0.bn.num_batches_tracked) == nbt + 2
    # an optimizer without found_inf: redone inside the step
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    with torch.no_grad():
        model.bn0.bn.weight.fill_(1e7)
    sgd = torch.optim.SGD(model.parameters(), lr=1e-6)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    loss, _ = train.train_step(model, sgd, *batch)
    assert model.train_range_fallbacks == 1 and np.isfinite(float(loss))
    assert int(model.bn0.bn.num_batches_tracked) == 1
    assert any(not torch.equal(before[k], p.detach()) for k, p in model.named_parameters())


def test_batchnorm_training_kernels_match_torch(cuda, built_lib):
    """MinkowskiBatchNorm in train mode (HIP statistics / apply / backward) vs torch.nn.BatchNorm1d on the CPU,
    including the running-statistics update and a strided feature view."""
    rng = np.random.default_rng(0)
    for n, c in ((5000, 96), (333, 32), (20000, 256)):
        x = (rng.normal(0.3, 2.0, (n, c)) * rng.uniform(0.5, 2, c)).astype(np.float32)
        gy = rng.normal(0, 1, (n, c)).astype(np.float32)
        ref = torch.nn.BatchNorm1d(c)
        with torch.no_grad():
            ref.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
            ref.bias.copy_(torch.from_numpy(rng.normal(0, 0.2, c).astype(np.float32)))
        mine = ME.MinkowskiBatchNorm(c)
        mine.bn.load_state_dict(ref.state_dict())
        mine = mine.cuda().train()
        ref.train()
        xr = torch.from_numpy(x).requires_grad_(True)
        yr = ref(xr)
        (yr * torch.from_numpy(gy)).sum().backward()
        xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
        coords = torch.cat([torch.zeros((n, 1), dtype=torch.int32), torch.arange(n, dtype=torch.int32)[:, None],
                            torch.zeros((n, 2), dtype=torch.int32)], 1)
        st = ME.SparseTensor(xd, coords, device="cuda")
        yd = mine(st).F
        (yd * torch.from_numpy(gy).to(cuda)).sum().backward()
        assert rel_err(yd.detach().cpu().numpy(), yr.detach().numpy()) < 1e-5
        assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-4
        assert rel_err(mine.bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-4
        assert rel_err(mine.bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy()) < 1e-4
        assert rel_err(mine.bn.running_mean.cpu().numpy(), ref.running_mean.numpy()) < 1e-5
        assert rel_err(mine.bn.running_var.cpu().numpy(), ref.running_var.numpy()) < 1e-5
        assert int(mine.bn.num_batches_tracked) == 1

        # BasicBlock's tail in one pass: relu(bn(x) + residual)
        res = rng.normal(0, 1, (n, c)).astype(np.float32)
        ref.zero_grad(); mine.zero_grad()
        xr = torch.from_numpy(x).requires_grad_(True)
        rr = torch.from_numpy(res).requires_grad_(True)
        yr = torch.relu(ref(xr) + rr)
        (yr * torch.from_numpy(gy)).sum().backward()
        xd = torch.from_numpy(x).to(cuda).requires_grad_(True)
        rd = torch.from_numpy(res).to(cuda).requires_grad_(True)
        yd = mine.forward_fused(ME.SparseTensor(xd, coords, device="cuda"), residual=rd, relu=True).F
        (yd * torch.from_numpy(gy).to(cuda)).sum().backward()
        assert rel_err(yd.detach().cpu().numpy(), yr.detach().numpy()) < 1e-5
        assert rel_err(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-4
        assert rel_err(rd.grad.cpu().numpy(), rr.grad.numpy()) < 1e-5
        assert rel_err(mine.bn.weight.grad.cpu().numpy(), ref.weight.grad.numpy()) < 1e-4
        assert rel_err(mine.bn.bias.grad.cpu().numpy(), ref.bias.grad.numpy()) < 1e-4


def _ddp_worker(rank, world_size, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world_size),
                      RANK=str(rank), LOCAL_RANK="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from canonicalvoting_amd import train
    from canonicalvoting_amd.minkunet import MinkUNet34C
    from canonicalvoting_amd.synth import make_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")            # both ranks share the one GPU of the test box: gloo, not RCCL

    def batch(seed):
        sc = make_scene(seed, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
        c = torch.cat([torch.zeros((1500, 1), dtype=torch.int32), torch.from_numpy(sc.coords).int()], 1).to(dev)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return c, t(sc.feats) * 2 - 1, t(sc.xyz_labels), t(sc.scale_labels), t(sc.class_labels)

    def grads(net, model, data):
        model.zero_grad(set_to_none=True)
        from canonicalvoting_amd import me as ME
        out = net(ME.SparseTensor(data[1], data[0], device=dev))
        loss, _ = train.joint_loss(out.F, data[2], data[3], data[4])
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    ddp = train.make_ddp(model, dev)
    g = grads(ddp, model, batch(40 + rank))                       # every rank its own scene
    if rank == 0:
        torch.manual_seed(0)
        ref_model = MinkUNet34C(3, 64).cuda().train()
        ref = [grads(ref_model, ref_model, batch(40 + r)) for r in range(world_size)]
        worst = 0.0
        for n in g:
            mean = sum(r[n] for r in ref) / world_size
            worst = max(worst, float((g[n] - mean).abs().max() / (mean.abs().max() + 1e-6)))
        q.put(("rank0", worst))
    names = ["final.kernel", "conv0p1s1.kernel", "block4.5.conv2.kernel"]
    q.put(("sig%d" % rank, [float(g[n].double().sum()) for n in names]))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradient_allreduce_two_ranks(cuda, built_lib):
    """BASELINE configs 3-4: scene-parallel DDP.  Two ranks (gloo, sharing this box's one GPU) each run the HIP
    forward/backward on their own scene through train.make_ddp; the reduced gradients are identical on both ranks and
    equal the mean of the two single-scene gradients (custom autograd Functions fire DDP's hooks correctly)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(3))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got["rank0"] < 1e-4, got
    np.testing.assert_allclose(got["sig0"], got["sig1"], rtol=1e-6)


def test_training_then_eval_uses_current_weights(cuda, built_lib):
    """fused optimizers update parameters without bumping tensor versions: the training path must not reuse packed
    weights of the previous step, and switching to eval mode must rebuild every derived cache (folded BatchNorm,
    packed weights, the C program)."""
    from canonicalvoting_amd import train
    from canonicalvoting_amd.synth import make_scene
    sc = make_scene(50, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    coords = torch.cat([torch.zeros((1500, 1), dtype=torch.int32), torch.from_numpy(sc.coords).int()], 1).to(cuda)
    feats, xyz, scale, cls = t(sc.feats) * 2 - 1, t(sc.xyz_labels), t(sc.scale_labels), t(sc.class_labels)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    opt = train.make_optimizer(model)                       # fused Adam on the GPU
    losses = [float(train.train_step(model, opt, coords, feats, xyz, scale, cls)[0]) for _ in range(6)]
    assert losses[-1] < 0.7 * losses[0], losses            # stale weights made the loss stall near its start value
    model.eval()
    with torch.no_grad():
        x = ME.SparseTensor(feats, coords, device=cuda)
        y_prog = model.program_forward(x).F
        y_fused = model.fused_forward(x).F
        y_mod = model.modular_forward(x).F
    assert float((y_prog - y_fused).abs().max()) < 2e-5 * max(1.0, float(y_fused.abs().max()))
    assert float((y_mod - y_fused).abs().max()) < 1e-4 * max(1.0, float(y_mod.abs().max()))
    # one more training step, then eval again: the eval caches must follow
    model.train()
    train.train_step(model, opt, coords, feats, xyz, scale, cls)
    model.eval()
    with torch.no_grad():
        y2 = model(ME.SparseTensor(feats, coords, device=cuda)).F
        y2_mod = model.modular_forward(ME.SparseTensor(feats, coords, device=cuda)).F
    assert float((y2 - y_prog).abs().max()) > 1e-6                      # the weights did move
    assert float((y2 - y2_mod).abs().max()) < 1e-4 * max(1.0, float(y2_mod.abs().max()))


def test_briefly_trained_network_stays_on_fp16_pairs_and_within_the_bar(cuda, built_lib):
    """VERDICT r2 "what's weak" 3: the fp16-pair eval path had only seen randomly initialised weights.  There is no
    checkpoint offline, so the network is TRAINED here - 150 `train_joint.py` steps (Adam 1e-3, three synthetic scenes per
    step, fresh scenes every 10 steps) - until weights, BatchNorm gains and running statistics are those of a network that
    has learnt something (loss down by more than half).  Then, on a scene it has not seen: the eval forward stays on the
    fp16 pairs (no range fallback), the largest activation between the convolutions is far inside the fp16 range, and the
    output is within the 1e-4 bar of the CPU oracle run on the trained state dict."""
    from canonicalvoting_amd import train
    from canonicalvoting_amd.synth import make_scene
    kw = dict(n_points=4000, res=0.05, room=(2.4, 1.4, 2.4), n_boxes=4, margin=0.6, box_scale=0.5)

    def batch(seed0):
        scenes = [make_scene(seed0 + b, **kw) for b in range(3)]
        coords = torch.cat([torch.cat([torch.full((len(s.coords), 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                            for b, s in enumerate(scenes)]).to(cuda)
        t = lambda name: torch.cat([torch.from_numpy(getattr(s, name)) for s in scenes]).to(cuda)
        return coords, t("feats") * 2 - 1, t("xyz_labels"), t("scale_labels"), t("class_labels")

    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().train()
    opt = train.make_optimizer(model, lr=1e-3)
    hist = []
    for step in range(150):
        if step % 10 == 0:
            data = batch(1000 + step)
        hist.append(float(train.train_step(model, opt, *data)[0]))
    assert np.isfinite(hist).all() and np.mean(hist[-10:]) < 0.5 * np.mean(hist[:3]), (hist[:3], hist[-10:])
    model.eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    sc = make_scene(77, **dict(kw, n_points=12000))
    coords = np.concatenate([np.zeros((len(sc.coords), 1), np.int64), sc.coords], 1)
    feats = (sc.feats * 2 - 1).astype(np.float32)
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        out = model(x).F.cpu().numpy()
    assert getattr(model, "range_fallbacks", 0) == 0
    so.relu_peaks = peaks = []
    try:
        ref = so.minkunet34c_forward(sd, coords, feats).numpy()
    finally:
        so.relu_peaks = None
    assert len(peaks) >= 50 and max(peaks) < 65504.0 / 16, max(peaks)     # fp16 range with a factor of 16 to spare
    print("trained-network activations: largest conv input %.3g over %d ReLUs; loss %.3f -> %.3f; max |out - oracle| %.2e"
          % (max(peaks), len(peaks), np.mean(hist[:3]), np.mean(hist[-10:]), np.abs(out - ref).max()))
    assert np.abs(out - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def _rccl_one_rank_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from canonicalvoting_amd import train
    from canonicalvoting_amd.minkunet import MinkUNet34C
    from canonicalvoting_amd.synth import make_scene
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sc = make_scene(41, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    c = torch.cat([torch.zeros((1500, 1), dtype=torch.int32), torch.from_numpy(sc.coords).int()], 1).to(dev)
    data = (c, t(sc.feats) * 2 - 1, t(sc.xyz_labels), t(sc.scale_labels), t(sc.class_labels))

    def two_steps(wrap):
        torch.manual_seed(0)
        model = MinkUNet34C(3, 64).cuda().train()
        net = wrap(model)
        opt = train.make_optimizer(model)
        losses = [float(train.train_step(net, opt, *data)[0]) for _ in range(2)]
        torch.cuda.synchronize()
        # gradients of the second step + the weights after it
        return losses, {n: p.grad.detach().clone() for n, p in model.named_parameters()}, \
            {n: p.detach().clone() for n, p in model.named_parameters()}

    plain = two_steps(lambda m: m)                       # before any process group exists: the non-DDP step
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)       # RCCL (backend "nccl" on ROCm)
    probe = torch.ones(4, device=dev)
    dist.all_reduce(probe)                               # librccl loaded, a communicator created, a collective run
    torch.cuda.synchronize()
    ddp = two_steps(lambda m: train.make_ddp(m, dev))
    backend = dist.get_backend()
    dist.barrier()
    dist.destroy_process_group()
    bad = [n for n in plain[1] if not torch.equal(plain[1][n], ddp[1][n])]
    badw = [n for n in plain[2] if not torch.equal(plain[2][n], ddp[2][n])]
    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
    q.put({"backend": backend, "losses": (plain[0], ddp[0]), "grad_diff": bad[:5], "n_grad_diff": len(bad),
           "n_weight_diff": len(badw), "rccl": sorted(set(loaded))[:2], "probe": float(probe.sum())})


def test_rccl_one_rank_ddp_equals_the_plain_step_bit_for_bit(cuda, built_lib):
    """RCCL at least once, on the one GPU there is.  A single-rank `nccl` (= RCCL) process group,
    train.make_ddp, two HIP training steps: librccl is mapped, DDP's bucket views and the reducer's hooks work with the
    custom autograd Functions (and with the side-stream weight gradients: a process group switches the late join off),
    and every gradient and every updated weight equals the non-DDP run bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(port, q))
    p.start()
    got = q.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert got["backend"] == "nccl" and got["rccl"], got
    assert got["probe"] == 4.0
    assert got["losses"][0] == got["losses"][1], got
    assert got["n_grad_diff"] == 0 and got["n_weight_diff"] == 0, got


def test_sorted_training_forward_equals_the_caller_order_forward(cuda, built_lib, monkeypatch):
    """MinkUNetBase.SORTED_TRAINING (CV_TRAIN_SORTED=1): the training forward on the spatially sorted twin of the coordinate
    set (stem through its map, `final` back through the inverse order, every map and mask order from one scene plan) is
    the same function: output rows in the caller's order, loss and every parameter gradient equal to the caller-order
    forward up to summation order."""
    from canonicalvoting_amd import train
    coords, feats = scene_coords(21, 9000, batch=2, small=False)
    rng = np.random.default_rng(3)
    perm = rng.permutation(len(coords))                       # the caller's rows in random order
    coords, feats = coords[perm], feats[perm]
    n = len(coords)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    xyz, scale = rng.normal(0, 0.5, (n, 3)).astype(np.float32), rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    cls = rng.integers(-1, 10, n).astype(np.int64)
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(MinkUNet34C, "SORTED_TRAINING", mode)
        torch.manual_seed(4)
        model = MinkUNet34C(3, 64).cuda().train()
        out = model(ME.SparseTensor(t(feats), t(coords).int(), device=cuda)).F
        loss = train.joint_loss(out, t(xyz), t(scale), t(cls))[0]
        loss.backward()
        res[mode] = (out.detach(), float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters()},
                     {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k})
    a, b = res[False], res[True]
    assert float((a[0] - b[0]).abs().max()) < 1e-4 * max(1.0, float(a[0].abs().max()))
    assert abs(a[1] - b[1]) < 1e-5 * max(1.0, abs(a[1]))
    # two evaluations with different summation orders pick different branches of ReLUs whose pre-activation is within
    # rounding of zero; a flipped element changes a gradient element by its whole value (the oracle's own fp32 autograd
    # sits 4e-2 from its fp64 run on the worst parameter, profiles/r3/relu_flip_probe.txt): the bulk has to agree
    errs = sorted(float((a[2][k] - b[2][k]).abs().max() / (a[2][k].abs().max() + 1e-12)) for k in a[2])
    assert errs[len(errs) // 2] < 2e-3 and errs[-1] < 0.1, (errs[len(errs) // 2], errs[-1])
    ga = torch.cat([a[2][k].flatten().double() for k in a[2]])
    gb = torch.cat([b[2][k].flatten().double() for k in a[2]])
    assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > 0.9999
    assert all(float((a[3][k] - b[3][k]).abs().max()) < 1e-5 for k in a[3])


def test_batched_weight_pack_equals_the_per_layer_packs(cuda, built_lib):
    """cv_sp_pack_weights_h2_batch_f32 against cv_sp_pack_weights_h2_f32 / cv_sp_pack_weights_t_f32, job by job: same fp16-pair
    words for a set of shapes of the network (1x1, k2s2, 3x3x3; forward and transposed; different scales), in one launch."""
    import ctypes
    from canonicalvoting_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    shapes = [(27, 96, 96, 0, 3), (27, 96, 96, 1, 3), (8, 64, 128, 0, -2), (8, 64, 128, 1, -2), (1, 128, 96, 0, 0), (27, 32, 64, 1, 7),
              (27, 256, 256, 0, 5)]
    ws = [torch.randn((K, cin, cout), generator=g).mul_(0.05).to(cuda) for K, cin, cout, _, _ in shapes]
    jobs = (_lib.PackJob * len(shapes))()
    outs, want = [], []
    st = torch.cuda.current_stream().cuda_stream
    for j, w, (K, cin, cout, trans, k) in zip(jobs, ws, shapes):
        o = torch.zeros(2 * w.numel(), dtype=torch.int16, device=cuda)
        outs.append(o)
        # trans: the packed tensor of the TRANSPOSED convolution of the forward kernel w[K][cin][cout] (Cin' = cout, Cout' = cin)
        j.w, j.wp, j.K, j.trans, j.scale_log2 = w.data_ptr(), o.data_ptr(), K, trans, k
        j.cin, j.cout = (cout, cin) if trans else (cin, cout)
        ref = torch.zeros_like(o)
        if trans:
            _lib.check(L.cv_sp_pack_weights_t_f32(w.data_ptr(), K, cin, cout, 2, k, ref.data_ptr(), st), "pack_t")
        else:
            _lib.check(L.cv_sp_pack_weights_h2_f32(w.data_ptr(), K, cin, cout, None, k, ref.data_ptr(), st), "pack_h2")
        want.append(ref)
    d_jobs = torch.empty(len(shapes) * ctypes.sizeof(_lib.PackJob), dtype=torch.uint8, device=cuda)
    _lib.check(L.cv_sp_pack_weights_h2_batch_f32(jobs, len(shapes), d_jobs.data_ptr(), st), "batch")
    torch.cuda.synchronize()
    for o, r, s in zip(outs, want, shapes):
        assert torch.equal(o, r), s
    jobs[0].cin = 48                                                     # Cin % 32 != 0 is refused, with a message
    assert L.cv_sp_pack_weights_h2_batch_f32(jobs, len(shapes), d_jobs.data_ptr(), st) == -22 and b"pack job 0" in L.cv_last_error()
    assert L.cv_sp_pack_weights_h2_batch_f32(None, 0, None, st) == 0


def test_guarded_multi_tensor_copy(cuda, built_lib):
    """cv_sp_copy_unless_flag (ME.copy_unless_flag): many small tensors of mixed dtypes in ceil(n / 96) launches; nothing is
    copied while the device-visible flag is non-zero (what keeps train.train_step's BatchNorm snapshot at the state before the
    first flagged step)."""
    g = torch.Generator().manual_seed(4)
    srcs = [torch.randn(int(n), generator=g).to(cuda) for n in torch.randint(1, 700, (150,), generator=g)]
    srcs += [torch.arange(5, dtype=torch.int64, device=cuda), torch.tensor(7, dtype=torch.int64, device=cuda)]
    dsts = [torch.zeros_like(s) for s in srcs]
    flag = ME.range_flag(torch.device(cuda))
    flag.zero_()
    ME.copy_unless_flag(srcs, dsts, torch.device(cuda), flag)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(srcs, dsts))
    old = [d.clone() for d in dsts]
    for s in srcs:
        s.add_(1)
    flag.fill_(1)
    try:
        ME.copy_unless_flag(srcs, dsts, torch.device(cuda), flag)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(old, dsts))               # guarded: the old values stay
        assert not any(torch.equal(a, b) for a, b in zip(srcs, dsts))
    finally:
        flag.zero_()
    ME.copy_unless_flag(srcs, dsts, torch.device(cuda))                        # no flag: always
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(srcs, dsts))
    ME.copy_unless_flag([], [], torch.device(cuda), flag)
