"""Pins the CPU vote oracle (oracle/hv_oracle.c): independent numpy restatement, analytic
known-answer tests, the fp32 grid-shape pitfall vectors and committed goldens.
The reference ships no test for this op, so these ARE the pin (parity otherwise unpinned)."""
import os

import numpy as np
import pytest

import oracle
from oracle import hv_numpy
from canonicalvoting_amd.synth import make_scene, synth_predictions

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def small_scene(seed, n=600, num_rots=24):
    sc = make_scene(seed, n_points=n, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5,
                    box_scale=0.4)
    xyz, scale, prob, cls = synth_predictions(sc)
    return sc, xyz, scale, prob, cls, num_rots


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_c_oracle_matches_numpy_restatement(seed):
    sc, xyz, scale, prob, _, R = small_scene(seed)
    g_obj, g_rot, g_scale, vin = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, R, return_vin=True)
    n_obj, n_rot, n_scale, nvin = hv_numpy.hv_forward(sc.points, xyz, scale, prob, sc.res, R)
    assert g_obj.shape == n_obj.shape
    assert vin == nvin                      # in-bounds decisions are bit-exact
    # same per-vote fp32 contributions, fp32-sequential vs float64 accumulation
    np.testing.assert_allclose(g_obj, n_obj, rtol=2e-5, atol=1e-5)
    live = n_obj > 1e-3                     # quotients of near-empty cells amplify rounding
    np.testing.assert_allclose(g_rot[live], n_rot[live], rtol=0, atol=1e-4)
    np.testing.assert_allclose(g_scale[live], n_scale[live], rtol=0, atol=1e-4)
    # exactly-empty cells stay exactly empty in both
    assert np.array_equal(g_obj == 0, n_obj == 0)


def test_grid_dims_fp32_pitfall():
    """coords are k*0.03f, so (max-min)/0.03f is often n - eps: the dim is n, not n+1
    (hv_cuda_kernel.cu:131-134; SURVEY.md 8a V1, vectors lo=-150)."""
    res = np.float32(0.03)
    for n in (2, 4, 11, 13, 248, 285):
        pts = np.array([[-150, 0, 0], [-150 + n, 1, 1]], np.float32) * res
        _, _, dims = oracle.grid_geometry(pts, float(res))
        _, _, ndims = hv_numpy.grid_dims(pts, res)
        assert dims[0] == n, (n, dims)
        assert ndims[0] == n
    # and an extent where the quotient is exact: dim n+1
    pts = np.array([[0, 0, 0], [8, 1, 1]], np.float32) * res
    _, _, dims = oracle.grid_geometry(pts, float(res))
    assert dims[0] == 9


def _anchored(points, xyz, scale, obj, lo, hi):
    """adds two zero-weight points that fix the grid's bounding box"""
    pts = np.concatenate([points, [lo, hi]]).astype(np.float32)
    xyz = np.concatenate([xyz, np.zeros((2, 3))]).astype(np.float32)
    scale = np.concatenate([scale, np.ones((2, 3))]).astype(np.float32)
    obj = np.concatenate([obj, [0, 0]]).astype(np.float32)
    return pts, xyz, scale, obj


def test_kat_point_on_node_with_zero_offset():
    """xyz = 0 => every rotation votes exactly on the point's own node: grid_obj[node] = R*obj,
    all other cells 0 (SURVEY.md 8c KAT 1).  res = 0.25 makes every quotient exact."""
    res, R = 0.25, 12
    pts, xyz, scale, obj = _anchored([[1.0, 0.5, 0.75]], [[0, 0, 0]], [[0.3, 0.2, 0.4]], [0.5],
                                     [0, 0, 0], [2, 2, 2])
    g_obj, g_rot, g_scale = oracle.hv_forward(pts, xyz, scale, obj, res, R)
    assert g_obj.shape == (9, 9, 9)
    assert g_obj[4, 2, 3] == np.float32(R * 0.5)
    assert g_obj.sum() == np.float32(R * 0.5)
    # scale channel = scale label, rot channel = mean of (cos, sin) over a full turn ~ 0
    np.testing.assert_allclose(g_scale[4, 2, 3], [0.3, 0.2, 0.4], rtol=1e-6)
    np.testing.assert_allclose(g_rot[4, 2, 3], [0, 0], atol=1e-6)


def test_kat_ring():
    """corr = (a,0,0) => votes lie on a ring of radius a in the plane y = const and the
    trilinear weights of each vote sum to obj: sum(grid_obj) = R*obj (KAT 2)."""
    res, R, a = 0.125, 40, 0.6
    pts, xyz, scale, obj = _anchored([[1.0, 1.0, 1.0]], [[1, 0, 0]], [[a, 1, 1]], [0.8],
                                     [0, 0, 0], [2, 2, 2])
    g_obj, _, _, vin = oracle.hv_forward(pts, xyz, scale, obj, res, R, return_vin=True)
    assert vin == 2 * R                     # + the zero-weight anchor at the origin (in bounds)
    np.testing.assert_allclose(g_obj.sum(dtype=np.float64), R * 0.8, rtol=1e-6)
    ys = np.nonzero(g_obj.sum((0, 2)))[0]
    assert list(ys) == [8]                  # y/res = 8 exactly: upper trilinear weight is 0
    xs, zs = np.nonzero(g_obj[:, 8, :])
    rad = np.hypot(xs * res - 1.0, zs * res - 1.0)
    assert rad.min() > a - 2 * res and rad.max() < a + 2 * res


def test_kat_box_peak_orientation_scale():
    """Exact LCC labels of a box with yaw t0 = k*2pi/R => peak at the centre cell,
    atan2(grid_rot) = t0, grid_scale = half extents (KAT 3)."""
    res, R, k = 0.05, 60, 7
    t0 = np.float32(k * (np.float32(2 * 3.141592654) / np.float32(R)))
    half = np.array([0.4, 0.3, 0.25])
    ctr = np.array([1.5, 0.5, 1.5])
    rng = np.random.default_rng(0)
    lcc = rng.uniform(-1, 1, (400, 3))
    lcc[np.arange(400), rng.integers(0, 3, 400)] = rng.choice([-1, 1], 400)   # on the faces
    c, s = np.cos(t0), np.sin(t0)
    Rm = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    world = ctr[None] + (lcc * half[None]) @ Rm.T
    pts, xyz, scale, obj = _anchored(world, lcc, np.tile(half, (400, 1)), np.ones(400),
                                     [0, 0, 0], [3, 1, 3])
    g_obj, g_rot, g_scale = oracle.hv_forward(pts, xyz, scale, obj, res, R)
    peak = np.unravel_index(np.argmax(g_obj), g_obj.shape)
    assert np.all(np.abs(np.array(peak) * res - ctr) <= res)
    assert g_obj[peak] > 100
    ang = np.arctan2(g_rot[peak][1], g_rot[peak][0])
    assert abs(((ang - t0) + np.pi) % (2 * np.pi) - np.pi) < 0.15
    np.testing.assert_allclose(g_scale[peak], half, rtol=1e-3)


def test_backward_matches_autograd_of_restatement():
    """Vote backward vs torch autograd of an independent float64 restatement of grid_obj.
    d_obj is the true gradient; d_xyz / d_scale are the true gradients times res because the
    reference omits the 1/res chain-rule factor (hv_cuda_kernel.cu:219-243 vs :198)."""
    import torch
    sc, xyz, scale, prob, _, R = small_scene(4, n=300, num_rots=16)
    pts = sc.points
    res = sc.res
    corner, _, dims = oracle.grid_geometry(pts, res)
    rng = np.random.default_rng(1)
    grad = rng.normal(0, 1, dims).astype(np.float32)
    d_xyz, d_scale, d_obj = oracle.hv_backward(grad, pts, xyz, scale, prob, res, R)

    ct, st = hv_numpy.rot_table(R)
    P = torch.tensor(pts, dtype=torch.float64)
    X = torch.tensor(xyz, dtype=torch.float64, requires_grad=True)
    S = torch.tensor(scale, dtype=torch.float64, requires_grad=True)
    O = torch.tensor(prob, dtype=torch.float64, requires_grad=True)
    G = torch.tensor(grad, dtype=torch.float64)
    C, Sn = torch.tensor(ct, dtype=torch.float64), torch.tensor(st, dtype=torch.float64)
    corr = X * S
    ox = -C[None] * corr[:, 0:1] + Sn[None] * corr[:, 2:3]
    oy = (-corr[:, 1:2]).expand_as(ox)
    oz = -Sn[None] * corr[:, 0:1] - C[None] * corr[:, 2:3]
    cr = torch.tensor(corner, dtype=torch.float64)
    g = [(P[:, k:k + 1] + o - cr[k]) / float(np.float32(res)) for k, o in enumerate((ox, oy, oz))]
    ok = (g[0] >= 0) & (g[1] >= 0) & (g[2] >= 0) & (g[0] < dims[0] - 1) & (g[1] < dims[1] - 1) & (g[2] < dims[2] - 1)
    fl = [torch.floor(a).long().clamp(0, d - 2) for a, d in zip(g, dims)]
    fr = [a - torch.floor(a) for a in g]
    total = 0
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                w = (fr[0] if bx else 1 - fr[0]) * (fr[1] if by else 1 - fr[1]) * (fr[2] if bz else 1 - fr[2])
                val = G[fl[0] + bx, fl[1] + by, fl[2] + bz]
                total = total + (w * O[:, None] * val * ok).sum()
    total.backward()
    np.testing.assert_allclose(d_obj, O.grad.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(d_xyz, X.grad.numpy() * res, rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(d_scale, S.grad.numpy() * res, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name", ["vote_512", "vote_2k"])
def test_oracle_reproduces_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    res = float(z["res"])
    pts = (z["coords"].astype(np.float32) * np.float32(res)).astype(np.float32)
    g_obj, g_rot, g_scale, vin = oracle.hv_forward(pts, z["xyz"], z["scale"], z["prob"], res,
                                                   int(z["num_rots"]), return_vin=True)
    assert list(g_obj.shape) == list(z["dims"])
    assert vin == int(z["v_in"])
    assert np.array_equal(g_obj, z["grid_obj"])          # the oracle is deterministic: bitwise
    if z["grid_rot"].size:
        assert np.array_equal(g_rot, z["grid_rot"])
        assert np.array_equal(g_scale, z["grid_scale"])
    np.testing.assert_allclose(g_rot.astype(np.float64).sum((0, 1, 2)), z["rot_sum"], rtol=1e-12)
    np.testing.assert_allclose(g_scale.astype(np.float64).sum((0, 1, 2)), z["scale_sum"], rtol=1e-12)
    d_xyz, d_scale, d_obj = oracle.hv_backward(z["grad"], pts, z["xyz"], z["scale"], z["prob"], res,
                                               int(z["num_rots"]))
    assert np.array_equal(d_xyz, z["d_xyz"]) and np.array_equal(d_scale, z["d_scale"])
    assert np.array_equal(d_obj, z["d_obj"])
    corner, _, _ = oracle.grid_geometry(pts, res)
    dec = oracle.decode(g_obj, g_rot, g_scale, corner, res, pts, z["xyz"], z["prob"], z["cls"],
                        oracle.DecodeParams.default(thresh_high=float(z["dec_thresh_high"])))
    assert np.array_equal(dec["cand_idx"], z["dec_cand"])
    assert np.array_equal(dec["verdict"], z["dec_verdict"])
    assert np.array_equal(dec["boxes"], z["dec_boxes"])
    assert np.array_equal(dec["classes"], z["dec_classes"])


def test_proposal_oracle_matches_reference_lines_executed_on_cpu():
    """oracle/proposal_oracle.py against sunrgbd/brnetcanon.py:104-162 exec()'d on CPU torch over the oracle's vote
    grids with the multinomial draws recorded (tests/golden/make_proposal_golden.py)"""
    import os
    import oracle
    from oracle import proposal_oracle as po
    from tests.golden.make_proposal_golden import make_inputs, RES, ROTS, NPROP
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "proposal_ref.npz"))
    pts, xyz, scale, prob, corners, votes = make_inputs(int(z["seed"]))
    g = oracle.hv_forward(pts, xyz, scale, prob, RES, ROTS, corners=corners)
    cand, scales, _, used = po.sample_proposals(g[0], g[2], corners[0], RES, votes, list(z["draws"]), NPROP)
    assert used == len(z["draws"])
    np.testing.assert_allclose(cand, z["candidates"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(scales, z["scales"])
    assert float(np.abs(z["probs"]).max()) == 0.0


def test_contribution_counts_agree_with_the_c_oracle():
    """hv_numpy.contribution_counts (the n_c of the quotient-grid bound in tests/test_vote_gpu.py): eight per in-bounds
    vote of the C oracle, every touched cell has some, chunking does not change it"""
    sc = make_scene(4, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
    xyz, scale, prob, _ = synth_predictions(sc)
    g = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, 36, return_vin=True)
    cnt = hv_numpy.contribution_counts(sc.points, xyz, scale, sc.res, 36)
    assert cnt.shape == g[0].shape and int(cnt.sum()) == 8 * g[3]
    assert (cnt[g[0] != 0] > 0).all()
    assert np.array_equal(cnt, hv_numpy.contribution_counts(sc.points, xyz, scale, sc.res, 36, chunk=97))
