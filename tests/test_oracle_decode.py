"""Known-answer tests for the CPU decode oracle (oracle/decode_oracle.c), the restatement of
eval_joint.py:195-280 + utils/calc_map.py:6-21."""
import numpy as np
import pytest

import oracle


def planted(shape=(40, 20, 40), res=0.05):
    X, Y, Z = shape
    g_obj = np.zeros(shape, np.float32)
    g_rot = np.zeros(shape + (2,), np.float32)
    g_rot[..., 0] = 1.0
    g_scale = np.full(shape + (3,), 0.2, np.float32)
    return g_obj, g_rot, g_scale


def box_points(center, half, yaw, n, rng, prob=0.9, noise=0.0):
    lcc = rng.uniform(-0.95, 0.95, (n, 3))
    c, s = np.cos(yaw), np.sin(yaw)
    Rm = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    pts = center[None] + (lcc * half[None]) @ Rm.T
    xyz = lcc + rng.normal(0, noise, lcc.shape) if noise else lcc
    return pts.astype(np.float32), xyz.astype(np.float32), np.full(n, prob, np.float32)


def test_tie_break_threshold_and_rejections():
    res = 0.05
    rng = np.random.default_rng(0)
    g_obj, g_rot, g_scale = planted()
    corner = np.zeros(3, np.float32)
    # A and B tie at 100: lowest flat index (A) must be examined first (torch.argmax).
    A, B, C, D, E = (10, 10, 10), (10, 10, 30), (30, 10, 10), (30, 10, 30), (20, 4, 20)
    g_obj[A] = 100; g_obj[B] = 100; g_obj[C] = 90; g_obj[D] = 80; g_obj[E] = 59.9   # E < thresh_high
    yawB = 0.7
    g_rot[B] = [np.cos(yawB), np.sin(yawB)]
    half = np.array([0.2, 0.2, 0.2])
    pa, xa, qa = box_points(np.array(A) * res, half, 0.0, 200, rng)            # accepted
    pb, xb, qb = box_points(np.array(B) * res, half, yawB, 150, rng)           # accepted (rotated)
    pc, xc, qc = box_points(np.array(C) * res, half, 0.0, 100, rng, prob=0.1)  # too few confident
    pd, xd, qd = box_points(np.array(D) * res, half, 0.0, 100, rng)
    xd = -xd                                                                   # LCC error > 0.3
    pts = np.concatenate([pa, pb, pc, pd]); xyz = np.concatenate([xa, xb, xc, xd])
    prob = np.concatenate([qa, qb, qc, qd])
    cls = np.concatenate([np.full(200, 3), np.r_[np.full(70, 5), np.full(80, 2)], np.full(100, 1),
                          np.full(100, 1)]).astype(np.int32)
    out = oracle.decode(g_obj, g_rot, g_scale, corner, res, pts, xyz, prob, cls)
    flat = lambda c: (c[0] * 20 + c[1]) * 40 + c[2]
    assert list(out["cand_idx"]) == [flat(A), flat(B), flat(C), flat(D)]       # E never examined
    assert list(out["verdict"]) == [0, 0, 1, 2]
    assert list(out["classes"]) == [3, 2]                                      # mode; 80 > 70
    np.testing.assert_allclose(out["scores"], [0.9, 0.9])
    # box A: axis-aligned +-0.2 around the cell centre, corner order of bbox_raw (eval_joint.py:203)
    raw = np.array([[1, 1, 1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1],
                    [-1, -1, -1], [-1, -1, 1]], np.float32)
    np.testing.assert_allclose(out["boxes"][0], raw * 0.2 + np.array(A) * res, atol=1e-6)
    # suppression: +-2 cube and the inside of the box are zero, E untouched
    after = out["grid_obj_after"]
    assert after[A] == 0 and after[B] == 0 and after[C] == 0 and after[D] == 0
    assert after[E] == np.float32(59.9)


def test_elimination_cube_variants_and_box_suppression():
    res = 0.05
    g_obj, g_rot, g_scale = planted()
    g_scale[...] = 0.01                      # tiny box: only the cube matters
    c = (20, 10, 20)
    g_obj[c] = 100
    g_obj[22, 10, 20] = 70                   # inside the joint cube (c+2), outside the separate one
    g_obj[25, 10, 20] = 65                   # outside both
    pts = np.zeros((1, 3), np.float32); xyz = pts.copy(); prob = np.zeros(1, np.float32)
    cls = np.zeros(1, np.int32)
    j = oracle.decode(g_obj, g_rot, g_scale, np.zeros(3, np.float32), res, pts, xyz, prob, cls)
    s = oracle.decode(g_obj, g_rot, g_scale, np.zeros(3, np.float32), res, pts, xyz, prob, cls,
                      oracle.DecodeParams.default(elim_hi_plus1=0))
    flat = lambda x, y, z: (x * 20 + y) * 40 + z
    assert list(j["cand_idx"]) == [flat(*c), flat(25, 10, 20)]                 # eval_joint.py:211
    assert list(s["cand_idx"]) == [flat(*c), flat(22, 10, 20), flat(25, 10, 20)]  # eval_separate.py:209
    # big box: every cell strictly inside the OBB is zeroed, boundary cells survive
    g_obj2, g_rot2, g_scale2 = planted()
    g_scale2[...] = 0.25                     # 5 cells half extent: |d| < 5 cells strictly
    g_obj2[c] = 100
    g_obj2[24, 10, 20] = 70                  # 4 cells away: inside -> suppressed
    g_obj2[25, 10, 20] = 66                  # exactly on the face: (5*0.05)/0.25 = 1 -> survives
    o = oracle.decode(g_obj2, g_rot2, g_scale2, np.zeros(3, np.float32), res, pts, xyz, prob, cls)
    assert list(o["cand_idx"]) == [flat(*c), flat(25, 10, 20)]


def test_iou_obb_known_answers():
    raw = np.array([[1, 1, 1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1],
                    [-1, -1, -1], [-1, -1, 1]], np.float32) * 0.5
    assert abs(oracle.iou_obb(raw, raw) - 1.0) < 1e-12
    assert oracle.iou_obb(raw, raw + np.array([3, 0, 0], np.float32)) == 0.0
    assert oracle.iou_obb(raw, raw + np.array([0, 2, 0], np.float32)) == 0.0      # y-disjoint
    t = np.pi / 4
    Rm = np.array([[np.cos(t), 0, -np.sin(t)], [0, 1, 0], [np.sin(t), 0, np.cos(t)]])
    rot = (raw @ Rm.T).astype(np.float32)
    octagon = 2 * (np.sqrt(2) - 1)                                             # unit squares at 45 deg
    assert abs(oracle.iou_obb(raw, rot) - octagon / (2 - octagon)) < 1e-6
    half_up = raw + np.array([0, 0.5, 0], np.float32)                          # half the height overlaps
    assert abs(oracle.iou_obb(raw, half_up) - 0.5 / 1.5) < 1e-6
    upside = raw[[4, 5, 6, 7, 0, 1, 2, 3]]                                     # calc_map.py:13 guard
    assert oracle.iou_obb(upside, raw) == 0


def test_nms_order_and_suppression():
    raw = np.array([[1, 1, 1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1],
                    [-1, -1, -1], [-1, -1, 1]], np.float32) * 0.5
    boxes = np.stack([raw, raw + np.float32([0.1, 0, 0]), raw + np.float32([5, 0, 0]),
                      raw + np.float32([5.05, 0, 0])])
    scores = np.array([0.5, 0.9, 0.7, 0.7], np.float32)
    assert oracle.nms(boxes, scores, 0.3) == [1, 3]      # stable ascending sort: later equal score wins
    assert oracle.nms(boxes, scores, 0.99) == [1, 3, 2, 0]
    assert oracle.nms(boxes[:0], scores[:0], 0.3) == []
    dets = oracle.nms_per_class(boxes, scores, np.array([2, 2, 0, 0]))
    assert [d[0] for d in dets] == [0, 2] and dets[0][2] == np.float32(0.7)


# ----------------------------------------------------------------------------------------------------------------
# Pinned by the reference itself: tests/golden/decode_ref_*.npz hold what eval_joint.py:173-190 (head split) and
# :195-263 (greedy decode loop) produced when those very lines were exec()'d on CPU torch tensors in the build
# container (tests/golden/make_decode_golden.py).  The oracle has to reproduce them.
@pytest.mark.parametrize("name", ["decode_ref_8k", "decode_ref_5k"])
def test_oracle_matches_reference_lines_executed_on_cpu(name):
    import os
    import torch
    from oracle import sparse_oracle as so
    from tests.golden.make_decode_golden import make_case
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    sc, F = make_case(int(z["seed"]), int(z["n"]), 1.0)
    res = float(z["res"])
    # head split (eval_joint.py:173-190)
    xyz, scale, prob, cls = [t.numpy() for t in so.head_joint_eval(torch.from_numpy(F))]
    assert np.array_equal(cls, z["class_pred"].astype(cls.dtype))
    np.testing.assert_array_equal(xyz, z["xyz_pred"])
    np.testing.assert_allclose(scale, z["scale_pred"], rtol=1e-6)
    np.testing.assert_allclose(prob, z["prob_pred"], rtol=1e-6, atol=1e-7)
    # decode loop (eval_joint.py:195-263) on the oracle's vote grids of the REFERENCE's head outputs
    pts = (sc.coords * np.float32(res)).astype(np.float32)
    g = oracle.hv_forward(pts, z["xyz_pred"], z["scale_pred"], z["prob_pred"], res, 120)
    corner, _, _ = oracle.grid_geometry(pts, res)
    th, tl, vr, el = z["consts"]
    d = oracle.decode(g[0], g[1], g[2], corner, res, pts, z["xyz_pred"], z["prob_pred"], z["class_pred"].astype(np.int32),
                      oracle.DecodeParams.default(thresh_high=th, thresh_low=tl, valid_ratio=vr, elimination=int(el)))
    assert len(d["boxes"]) == len(z["boxes"])
    assert list(d["classes"]) == list(z["classes"])
    np.testing.assert_allclose(d["boxes"], z["boxes"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(d["scores"], z["scores"], rtol=1e-6)
    zeroed = np.flatnonzero((g[0] != 0) & (d["grid_obj_after"] == 0))
    assert np.array_equal(zeroed, z["zeroed"].astype(zeroed.dtype))


def test_joint_loss_matches_reference_lines_executed_on_cpu():
    """train.joint_loss against train_joint.py:253-283 exec()'d on CPU torch (tests/golden/make_loss_golden.py):
    the three terms, their sum and the gradient w.r.t. the network output"""
    import os
    import torch
    from canonicalvoting_amd import train
    from tests.golden.make_loss_golden import make_inputs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_ref.npz"))
    F, labels, xyz, scale = make_inputs(int(z["seed"]), int(z["n"]))
    out = torch.from_numpy(F.copy()).requires_grad_(True)
    loss, parts = train.joint_loss(out, torch.from_numpy(xyz), torch.from_numpy(scale), torch.from_numpy(labels))
    loss.backward()
    for k in ("loss_xyz", "loss_scale", "loss_class"):
        assert abs(float(parts[k]) - float(z[k])) < 1e-6 * max(1.0, abs(float(z[k]))), k
    assert abs(float(loss) - float(z["loss"])) < 1e-6 * max(1.0, abs(float(z["loss"])))
    np.testing.assert_allclose(out.grad.numpy(), z["grad"], rtol=1e-5, atol=1e-8)


def test_joint_loss_never_touches_background_rows():
    """train_joint.py:262-272 index the predictions with the object mask: a non-finite prediction in a background row reaches
    neither the loss nor a gradient.  The host-wait-free form of train.joint_loss selects (torch.where) instead of indexing -
    and must not multiply by zero (inf * 0 = NaN): ADVICE r5."""
    import os
    import torch
    from canonicalvoting_amd import train
    from tests.golden.make_loss_golden import make_inputs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_ref.npz"))
    F, labels, xyz, scale = make_inputs(int(z["seed"]), int(z["n"]))
    bg = np.nonzero((labels == 9) | (labels < 0))[0]
    assert len(bg) >= 3
    bad = F.copy()
    bad[bg[0], :27] = np.inf              # xyz heads of a background row
    bad[bg[1], 27:54] = -np.inf           # scale heads
    bad[bg[2], :54] = np.nan
    out = torch.from_numpy(bad).requires_grad_(True)
    loss, parts = train.joint_loss(out, torch.from_numpy(xyz), torch.from_numpy(scale), torch.from_numpy(labels))
    loss.backward()
    for k in ("loss_xyz", "loss_scale", "loss_class"):
        assert abs(float(parts[k]) - float(z[k])) < 1e-6 * max(1.0, abs(float(z[k]))), k
    g = out.grad.numpy()
    assert np.isfinite(g).all()
    assert not g[bg[:3], :54].any()                      # no gradient into the regression heads of background rows
    np.testing.assert_allclose(g, z["grad"], rtol=1e-5, atol=1e-8)


def test_separate_loss_matches_reference_lines_executed_on_cpu():
    """train.separate_loss against train_separate.py:247-286 exec()'d on CPU torch over a two-scan batch of the mini
    dataset: objectness CE, log-scale MSE, minimum-over-symmetric-poses coordinate loss, sum, gradient.  The
    reference indexes the batch output with per-scan row numbers (:271); reference_indexing=True reproduces that,
    False is the repaired variant (differs as soon as the batch holds a second scan)."""
    import os
    import torch
    from canonicalvoting_amd import train
    from tests.golden.make_loss_golden import make_separate_inputs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loss_ref.npz"))
    batch, F = make_separate_inputs()
    out = torch.from_numpy(F.copy()).requires_grad_(True)
    loss, parts = train.separate_loss(out, batch[3], batch[4], batch[5])
    loss.backward()
    for k in ("loss_obj", "loss_xyz", "loss_scale"):
        assert abs(float(parts[k]) - float(z["sep_" + k])) < 1e-6 * max(1.0, abs(float(z["sep_" + k]))), k
    assert abs(float(loss) - float(z["sep_loss"])) < 1e-6
    np.testing.assert_allclose(out.grad.numpy(), z["sep_grad"], rtol=1e-5, atol=1e-8)
    fixed, parts2 = train.separate_loss(torch.from_numpy(F), batch[3], batch[4], batch[5], coords4=batch[1],
                                        reference_indexing=False)
    assert abs(float(parts2["loss_xyz"]) - float(parts["loss_xyz"])) > 1e-4
    assert float(parts2["loss_scale"]) == float(parts["loss_scale"].detach())
    # a single-scan batch: both indexings coincide
    one = [batch[3][0]]
    n0 = int((batch[1][:, 0] == 0).sum())
    a, _ = train.separate_loss(torch.from_numpy(F[:n0]), one, batch[4][:n0], batch[5][:n0])
    b, _ = train.separate_loss(torch.from_numpy(F[:n0]), one, batch[4][:n0], batch[5][:n0], coords4=batch[1][:n0],
                               reference_indexing=False)
    assert float(a) == float(b)
