"""Parity at the sizes BASELINE.json quotes, on the exact launch path bench.py times.

  config 2   one 80 000-point scene: the C-program MinkUNet34C forward (4 mask groups, split-K sizing for 256 CUs,
             one-arena scene maps, C executor - everything rows >= 16 384 switches on) against the CPU oracle,
             head classes, vote + decode of the network's own predictions
  config 5   one 300 000-point SUN RGB-D shaped scene (9 x 3 x 9 m room, 40 boxes, grid ~302 x 102 x 302):
             coordinate sets / kernel maps exact at that hash occupancy, vote grid exact integer outputs,
             two separate 8-channel models on ONE SparseTensor (eval_separate.py:162-186)

Bars: integer outputs (coordinate sets, kernel maps, classes, grid shape, in-bounds vote count, touched-cell set,
candidates, verdicts, box count) exact; floats within 1e-4 (north_star), with the per-cell bound of the quotient grids
of tests/test_vote_gpu.py."""
import numpy as np
import pytest
import torch

import oracle
from oracle import sparse_oracle as so
from canonicalvoting_amd import decode, hv_cuda, pipeline
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene, synth_predictions
from tests.test_vote_gpu import assert_grids_close

pytestmark = pytest.mark.gpu
RES, R = 0.03, 120


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def decided_points(class_logits_ref, err):
    """points whose two argmaxes (over the 9 classes + background for the head select, over the 9 classes for the
    label: eval_joint.py:176-190) have a top-2 margin in the ORACLE's logits above twice the float error between the
    two networks - a tie inside the float tolerance has no defined winner"""
    ok = np.ones(len(class_logits_ref), bool)
    for logits in (class_logits_ref, class_logits_ref[:, :-1]):
        top2 = np.sort(logits, 1)[:, -2:]
        ok &= (top2[:, 1] - top2[:, 0]) > 2 * err
    return ok


# the two launch sizings bench.py runs under: one scene at a time (the library's defaults: its isolated pass) and seven
# scenes in flight (the TIMED region: split target 256, 12288 vote records per part, mask-sorted convolutions from 8192
# rows - the ts4 level of an 80k scene, 9 649 rows, then runs mask-sorted).  VERDICT r5 weak 2: every parity case at
# BASELINE's sizes runs under both.
POLICIES = [pytest.param(1, id="one-in-flight"), pytest.param(7, id="seven-in-flight")]


@pytest.fixture(scope="module")
def net80k():
    """bench.py's scene 0 and network, and the CPU oracle's forward of it (computed once for both launch policies)"""
    sc = make_scene(0, n_points=80000)
    coords4 = np.concatenate([np.zeros((80000, 1), np.int64), sc.coords], 1)
    feats = (sc.feats * 2 - 1).astype(np.float32)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().eval()          # bench.py's network (eval_joint.py:151)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = so.minkunet34c_forward(sd, coords4, feats).numpy()
    return sc, coords4, feats, model, ref


@pytest.mark.parametrize("in_flight", POLICIES)
def test_config2_80k_network_program_matches_oracle(cuda, built_lib, net80k, in_flight):
    sc, coords4, feats, model, ref = net80k
    policy = pipeline.policy_for_scenes_in_flight(in_flight)
    x = ME.SparseTensor(dev(feats, cuda), dev(coords4, cuda).int(), device=cuda)
    with torch.no_grad(), pipeline.scene_policy(policy):
        y = model(x, defer_check=True)                           # the call bench.py's run_step makes
        assert model.check_range(x, y) is y                      # no fp16-range fallback on this scene
        pred = pipeline.head_joint(y.F)
        plan = x.coordinate_manager.fused_fast()
    assert plan is not None and x.coordinate_manager.num_rows(1) == 80000 >= 16384      # the mask-group regime
    # levels 0 and 1 (80 000 / ~36 500 rows) are mask-sorted under both policies, level 2 (~9 600 rows) only with seven in flight
    assert [p is not None for p in plan.perm_ptrs[:5]] == [True, True, in_flight >= 4, False, False], plan.counts
    got = y.F.cpu().numpy()
    err = float(np.abs(got - ref).max())
    assert err < 1e-4 * max(1.0, float(np.abs(ref).max())), err
    rx, rs, rp, rc = [t.numpy() for t in so.head_joint_eval(torch.from_numpy(ref))]
    xyz, scale, prob, cls = [t.cpu().numpy() for t in pred]
    ok = decided_points(ref[:, 6 * 9:], err)
    assert (~ok).sum() < 80                                      # < 0.1 % of the points sit on a float-level tie
    assert np.array_equal(cls[ok], rc[ok])                       # head classes exact
    np.testing.assert_allclose(prob, rp, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(scale[ok], rs[ok], rtol=1e-4)
    np.testing.assert_allclose(xyz[ok], rx[ok], rtol=0, atol=1e-4)
    # vote + decode on the network's own (HIP) predictions vs the oracle on the same predictions
    pts = (sc.coords * RES).astype(np.float32)
    hv = HoughVoting(RES, R)
    with torch.no_grad(), pipeline.scene_policy(policy):
        grids = hv(dev(pts, cuda), *pred[:3])
    refg = oracle.hv_forward(pts, xyz, scale, prob, RES, R, return_vin=True)
    assert_grids_close([g.cpu().numpy() for g in grids], refg[:3], "80k network predictions",
                       inputs=(pts, xyz, scale, RES, R))
    corner, _, dims = oracle.grid_geometry(pts, RES)
    assert hv_cuda.count_votes(dev(pts, cuda), pred[0], pred[1], RES, R, corner, dims) == refg[3]


@pytest.fixture(scope="module")
def teacher80k():
    """bench.py's scene 0 with teacher predictions and the oracle's vote grids of it (once for both launch policies)"""
    sc = make_scene(0, n_points=80000)
    pred = synth_predictions(sc)
    ref = oracle.hv_forward(sc.points, pred[0], pred[1], pred[2], RES, R, return_vin=True)
    return sc, pred, ref


@pytest.mark.parametrize("in_flight", POLICIES)
def test_config2_80k_teacher_scene_end_to_end_matches_oracle(cuda, built_lib, teacher80k, in_flight):
    """the workload of bench.py's default line: vote + decode + NMS on teacher predictions of an 80k scene - the grids
    against the oracle's (every cell), the decode against the oracle's decode of the same grids"""
    sc, (xyz, scale, prob, cls), ref = teacher80k
    pts = sc.points
    hv = HoughVoting(RES, R)
    p, x, s, o, c = [dev(a, cuda) for a in (pts, xyz, scale, prob, cls)]
    with torch.no_grad(), pipeline.scene_policy(pipeline.policy_for_scenes_in_flight(in_flight)):
        grids = hv(p, x, s, o)
        g = [t.cpu().numpy() for t in grids]
        raw = decode.decode_boxes(*grids, p, x, o, c, RES)
    assert_grids_close(g, ref[:3], "80k teacher predictions, %d in flight" % in_flight, inputs=(pts, xyz, scale, RES, R))
    corner, _, _ = oracle.grid_geometry(pts, RES)
    d = oracle.decode(g[0], g[1], g[2], corner, RES, pts, xyz, prob, cls)
    assert np.array_equal(raw["cand_idx"], d["cand_idx"]) and np.array_equal(raw["verdict"], d["verdict"])
    assert len(raw["boxes"]) == len(d["boxes"]) >= 8 and list(raw["classes"]) == list(d["classes"])
    np.testing.assert_allclose(raw["boxes"], d["boxes"], rtol=0, atol=1e-5)
    dets = decode.nms_per_class(raw["boxes"], raw["scores"], raw["classes"])
    rdet = oracle.nms_per_class(d["boxes"], d["scores"], d["classes"])
    assert len(dets) == len(rdet)


def test_vote_grids_are_the_same_bits_for_every_part_size(cuda, built_lib, teacher80k):
    """cv_hv_set_part_records[_thread] / cv_scene_desc.vote_part_records: the records one workgroup of a hot (tile, plane)
    takes only decide how a plane's votes are spread over workgroups; the parts are added as 2^-36 fixed-point integers, so
    the three grids are torch.equal at 4096 (library), 12288 (bench.py's timed region) and 16384 records per part."""
    sc, (xyz, scale, prob, cls), _ = teacher80k
    hv = HoughVoting(RES, R)
    args = [dev(a, cuda) for a in (sc.points, xyz, scale, prob)]
    grids = {}
    for records in (4096, 12288, 16384):
        with torch.no_grad(), pipeline.scene_policy(pipeline.ScenePolicy(0, records, ME.masked_min_rows())):
            grids[records] = [g.clone() for g in hv(*args)]
    for records in (12288, 16384):
        for name, a, b in zip(("obj", "rot", "scale"), grids[4096], grids[records]):
            assert torch.equal(a, b), "grid_%s at %d records per part: %d cells differ" % (name, records, int((a != b).sum()))
    assert float(grids[4096][0].max()) > 60.0


@pytest.mark.parametrize("rows", [8192, 4096])
def test_mask_sorted_coarse_levels_stay_within_1e4_of_the_oracle(cuda, built_lib, net80k, rows):
    """masked_min_rows 8192 (bench.py's timed region: ts4, 9 649 rows, mask-sorted) and 4096: the network output within
    1e-4 of the oracle, through the one-call scene path (cv_scene_desc.masked_min_rows) and equal to the module path's
    output under the same policy bit for bit"""
    sc, coords4, feats, model, ref = net80k
    policy = pipeline.ScenePolicy(256, 12288, rows)
    c4, f = dev(coords4, cuda).int(), dev(feats, cuda)
    hv = HoughVoting(RES, R)
    keep = {}
    pipeline.detect_scene_c(model, hv, c4, f, RES, keep=keep, policy=policy)
    assert sum(r >= rows for r in keep["level_rows"]) >= 3
    got = keep["y"].cpu().numpy()
    err = float(np.abs(got - ref).max())
    assert err < 1e-4 * max(1.0, float(np.abs(ref).max())), err
    with torch.no_grad(), pipeline.scene_policy(policy):
        y = model(ME.SparseTensor(f, c4, device=cuda)).F
    assert torch.equal(y, keep["y"])


@pytest.fixture(scope="module")
def scene300k():
    sc = make_scene(3, n_points=300000, room=(9.0, 3.0, 9.0), n_boxes=40)
    assert len(sc.coords) == 300000
    return sc


def test_config5_300k_coordinate_sets_and_kernel_maps_exact(cuda, built_lib, scene300k):
    sc = scene300k
    coords4 = np.concatenate([np.zeros((len(sc.coords), 1), np.int64), sc.coords], 1)
    cm = ME.CoordinateManager(dev(coords4, cuda).int())
    ocm = so.CoordinateManager(coords4)
    for ts in (1, 2, 4, 8, 16):
        assert np.array_equal(cm.coords[ts].cpu().numpy(), ocm.coords[ts]), ts
    for k, ts, stride in ((5, 1, 1), (3, 1, 1), (3, 2, 1), (3, 4, 1), (3, 8, 1), (3, 16, 1), (2, 1, 2), (2, 2, 2),
                          (2, 4, 2), (2, 8, 2)):
        assert np.array_equal(cm.kernel_map(k, ts, stride).cpu().numpy(), ocm.map(k, ts, stride)), (k, ts, stride)


def test_config5_300k_vote_grid_matches_oracle(cuda, built_lib, scene300k):
    sc = scene300k
    xyz, scale, prob, cls = synth_predictions(sc)
    pts = sc.points
    ref = oracle.hv_forward(pts, xyz, scale, prob, RES, R, return_vin=True)
    assert ref[0].shape[0] > 290 and ref[0].shape[2] > 290          # the ~302 x 102 x 302 grid of config 5
    hv = HoughVoting(RES, R)
    p, x, s, o, c = [dev(a, cuda) for a in (pts, xyz, scale, prob, cls)]
    with torch.no_grad():
        grids = hv(p, x, s, o)
    assert_grids_close([g.cpu().numpy() for g in grids], ref[:3], "300k", inputs=(pts, xyz, scale, RES, R))
    corner, _, dims = oracle.grid_geometry(pts, RES)
    assert hv_cuda.count_votes(p, x, s, RES, R, corner, dims) == ref[3]
    raw = decode.decode_boxes(*grids, p, x, o, c, RES)
    g = [t.cpu().numpy() for t in hv(p, x, s, o)]
    d = oracle.decode(g[0], g[1], g[2], corner, RES, pts, xyz, prob, cls)
    assert np.array_equal(raw["cand_idx"], d["cand_idx"]) and np.array_equal(raw["verdict"], d["verdict"])
    assert len(raw["boxes"]) == len(d["boxes"]) and list(raw["classes"]) == list(d["classes"])
    assert len(d["boxes"]) >= 20                                      # most of the 40 planted boxes come back


def test_config5_300k_separate_heads_on_one_sparse_tensor(cuda, built_lib, scene300k):
    """eval_separate.py:162-186 at config-5 size: two 8-channel models share one SparseTensor / coordinate manager;
    the first is compared with the oracle network at full size"""
    sc = scene300k
    coords4 = np.concatenate([np.zeros((len(sc.coords), 1), np.int64), sc.coords], 1)
    feats = (sc.feats * 2 - 1).astype(np.float32)
    sds = [so.make_state_dict(3, 8, seed=50 + i) for i in range(2)]
    models = []
    for sd in sds:
        m = MinkUNet34C(3, 8)
        m.load_state_dict(sd)
        models.append(m.cuda().eval())
    x = ME.SparseTensor(dev(feats, cuda), dev(coords4, cuda).int(), device=cuda)
    with torch.no_grad():
        ys = [m(x).F for m in models]
    plan = x.coordinate_manager.fused_plan()
    assert x.coordinate_manager.fused_plan() is plan                   # maps built once for both models
    ref = so.minkunet34c_forward(sds[0], coords4, feats)
    err = float((ys[0].cpu() - ref).abs().max())
    assert err < 1e-4 * max(1.0, float(ref.abs().max())), err
    # the same network under the launch sizing of a host with seven scenes in flight (bench.py --large): within the same bar of
    # the oracle, and another summation order than the library's policy (the policy did arrive)
    with torch.no_grad(), pipeline.scene_policy(pipeline.policy_for_scenes_in_flight(7)):
        y7 = models[0](ME.SparseTensor(dev(feats, cuda), dev(coords4, cuda).int(), device=cuda)).F
    err7 = float((y7.cpu() - ref).abs().max())
    assert err7 < 1e-4 * max(1.0, float(ref.abs().max())), err7
    assert not torch.equal(y7, ys[0])
    xyz, scale, prob = pipeline.head_separate(ys[0])
    rx, rs, rp = so.head_separate_eval(ys[0].cpu())
    np.testing.assert_allclose(xyz.cpu().numpy(), rx.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(scale.cpu().numpy(), rs.numpy(), rtol=1e-5)
    np.testing.assert_allclose(prob.cpu().numpy(), rp.numpy(), rtol=1e-5, atol=1e-6)
    assert torch.isfinite(ys[1]).all() and not torch.equal(ys[0], ys[1])


def _train_batch(cuda, B, n, seed0):
    scenes = [make_scene(seed0 + b, n_points=n) for b in range(B)]
    coords = np.concatenate([np.concatenate([np.full((n, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
    feats = np.concatenate([s.feats for s in scenes]).astype(np.float32) * 2 - 1
    labels = [np.concatenate([getattr(s, k) for s in scenes]) for k in ("xyz_labels", "scale_labels", "class_labels")]
    return coords, feats, labels


def test_config3_training_forward_and_loss_at_three_80k_scenes(cuda, built_lib):
    """config 3 at the size config/config.yaml:15 trains at (batch of 3 scans, 3 x 80k rows): the
    TRAINING-mode forward (batch-statistics BatchNorm over the 240k rows, mask-sorted groups, bf16-triple products) and
    the loss of train_joint.py:253-283 against the oracle's training-mode forward; the backward runs and is finite.
    (The oracle's autograd does not fit at this size - its gather-matmul-scatter keeps [pairs, C] intermediates of
    every layer; the gradients are checked at 3 x 20k rows below, where every large-size code path is already on.)"""
    from canonicalvoting_amd import train
    coords, feats, (xyz, scale, cls) = _train_batch(cuda, 3, 80000, 40)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    out = model(ME.SparseTensor(dev(feats, cuda), dev(coords, cuda).int(), device=cuda)).F
    loss = train.joint_loss(out, dev(xyz, cuda), dev(scale, cuda), dev(cls, cuda))[0]
    loss.backward()
    assert all(torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in model.parameters())
    with torch.no_grad():
        ref = so.minkunet34c_forward(sd, coords, feats, training=True)
        lo = train.joint_loss(ref, torch.from_numpy(xyz), torch.from_numpy(scale), torch.from_numpy(cls))[0]
    err = float((out.detach().cpu() - ref).abs().max())
    assert err < 1e-4 * max(1.0, float(ref.abs().max())), err
    assert abs(float(loss.detach()) - float(lo)) < 1e-4 * max(1.0, abs(float(lo)))


def _training_gradients_on_shared_relu_masks(cuda, n_points, seed0, separate=False):
    """every parameter gradient of one train_joint.py step on 3 x 20k rows against autograd through the CPU oracle in
    DOUBLE precision: at 60k rows the finest two levels run the mask-sorted groups (>= 16384 rows), the weight-gradient
    kernels their chunked plans and the coarse levels their split-K sizing - the code paths of the 3 x 80k step, at a
    size the oracle's autograd still fits.

    The comparison is made on the SAME activation pattern: the masks (y > 0) of the HIP forward's 55 ReLUs are handed to
    the oracle (sparse_oracle.relu_masks).  Two evaluations of this network do not agree to rounding on the gradients
    otherwise: a pre-activation within ~1e-6 of zero gets a different ReLU mask, which changes that gradient element by
    its whole value and a weight gradient (a sum over N rows of zero-mean terms) by ~1/sqrt(N) per flipped element - the
    oracle's own fp32 autograd sits 4e-2 (worst parameter) from its fp64 run at 3 x 5k rows and 2.5e-6 from it once
    both use one set of masks (profiles/r3/relu_flip_probe.txt).  A forced mask changes the forward only where
    |x| ~ 1e-6, so the loss still has to match."""
    from canonicalvoting_amd import train
    coords, feats, (xyz, scale, cls) = _train_batch(cuda, 3, n_points, seed0)
    if separate:
        # train_separate.py:247-287 on synthetic per-category labels: objectness = "point of an object", per scan three
        # models, each a set of object rows with its coordinates under two symmetry-equivalent poses (the loss takes the
        # minimum over the poses); reference_indexing=False so that every scan's rows are its own
        rng = np.random.default_rng(7)
        obj = (cls != 9).astype(np.int64)
        sc_lab = np.where(scale > 0, scale, 1.0).astype(np.float32)
        per_scan = []
        for b in range(3):
            rows_b = np.nonzero(obj[b * n_points:(b + 1) * n_points])[0]
            models = []
            for _ in range(3):
                rows = np.sort(rng.choice(rows_b, size=min(len(rows_b), 400), replace=False))
                x0 = xyz[b * n_points + rows]
                models.append((rows, [x0, x0 * np.array([-1, 1, -1], np.float32)]))
            per_scan.append(models)

        def loss_of(out, to):
            lab = [[(torch.from_numpy(r), [to(x) for x in xs]) for r, xs in ms] for ms in per_scan]
            return train.separate_loss(out, lab, to(sc_lab), torch.from_numpy(obj).to(out.device),
                                       coords4=torch.from_numpy(coords).to(out.device), reference_indexing=False)[0]
    else:
        def loss_of(out, to):
            return train.joint_loss(out, to(xyz), to(scale), torch.from_numpy(cls).to(out.device))[0]
    torch.manual_seed(1)
    model = MinkUNet34C(3, 8 if separate else 6 * 9 + 9 + 1).cuda().train()
    model.SORTED_TRAINING = False            # (the masks are compared row by row with the oracle's: the caller's order)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    masks = []
    fused = ME.MinkowskiBatchNorm.forward_fused

    def recording(self, x, residual=None, relu=False):
        y = fused(self, x, residual=residual, relu=relu)
        if relu:
            masks.append((y.F.detach() > 0).cpu())
        return y

    ME.MinkowskiBatchNorm.forward_fused = recording
    try:
        out = model(ME.SparseTensor(dev(feats, cuda), dev(coords, cuda).int(), device=cuda)).F
    finally:
        ME.MinkowskiBatchNorm.forward_fused = fused
    loss = loss_of(out, lambda a: dev(a, cuda))
    loss.backward()
    pnames = [k for k, _ in model.named_parameters()]
    dt = torch.float64
    sdo = {k: (v.clone().to(dt).requires_grad_(True) if k in pnames else v.clone()) for k, v in sd.items()}
    so.relu_masks = iter(masks)
    try:
        yo = so.minkunet34c_forward(sdo, coords, feats.astype(np.float64), training=True, dtype=dt)
        assert next(so.relu_masks, None) is None, "the oracle applied fewer ReLUs than the HIP forward"
    finally:
        so.relu_masks = None
    lo = loss_of(yo, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dt))
    lo.backward()
    assert len(masks) == 55
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-4 * max(1.0, abs(float(lo.detach())))
    errs = []
    for name, p in model.named_parameters():
        g, go = p.grad.double().cpu().numpy(), sdo[name].grad.numpy()
        errs.append((float(np.abs(g - go).max() / max(1e-12, np.abs(go).max())), name))
    errs.sort(reverse=True)
    print("%s: largest parameter-gradient errors at 3 x %d rows, same ReLU masks (max |d| / max |g|):"
          % ("train_separate.py loss" if separate else "train_joint.py loss", n_points),
          [(n, "%.2e" % e) for e, n in errs[:4]], "median %.2e" % errs[len(errs) // 2][0])
    assert errs[0][0] < 1e-4, errs[:6]


def test_config3_training_gradients_at_three_20k_scenes(cuda, built_lib):
    _training_gradients_on_shared_relu_masks(cuda, 20000, 60)


def test_config5_separate_training_gradients_at_three_8k_scenes(cuda, built_lib):
    """config 5's training side (train_separate.py: an 8-channel model per category, coordinate loss = minimum over the
    symmetry-equivalent poses): loss and all parameter gradients vs the fp64 oracle on shared ReLU masks.  3 x 8k rows
    keep the mask-sorted groups of the finest level (>= 16384 rows) in the path at a third of the oracle's time; at
    3 x 20k rows the same test gave 5.3e-6 worst / 2.4e-6 median (round 3)."""
    _training_gradients_on_shared_relu_masks(cuda, 8000, 70, separate=True)


def test_config3_training_gradients_at_three_80k_scenes(cuda, built_lib):
    """the same at the size config/config.yaml:15 trains at: 240k rows, all 249 parameter gradients.
    The fp64 autograd of the oracle keeps the [pairs, C] intermediates of every layer (~60 GB of host memory) and takes
    five minutes on the GPU box's cores, so the test is opt-in (CV_TEST_3X80K=1); its output of this round is committed as
    profiles/r3/train_gradients_3x80k.txt (worst parameter 6.5e-6, median 2.9e-6)."""
    import os
    if os.environ.get("CV_TEST_3X80K", "0") != "1":
        pytest.skip("opt-in: CV_TEST_3X80K=1 (five minutes of CPU autograd in fp64, ~60 GB of host memory)")
    mem = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    if mem < (128 << 30):
        pytest.skip("the oracle's fp64 autograd at 3 x 80k rows needs ~60 GB of host memory")
    _training_gradients_on_shared_relu_masks(cuda, 80000, 40)
