"""HIP decode (compaction + single-workgroup greedy + parallel back-projection) vs the CPU
oracle's sequential restatement of eval_joint.py:195-263.  Integer outputs (candidate cells,
verdicts, box count, classes) must be exact; boxes/scores are bit-identical by construction
(shared fp32 conventions), asserted to 1e-6."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from canonicalvoting_amd import decode
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions

pytestmark = pytest.mark.gpu


def t(cuda, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def compare(cuda, g_obj, g_rot, g_scale, corner, res, pts, xyz, prob, cls, **kw):
    okw = dict(kw)
    if "separate_variant" in okw:
        okw["elim_hi_plus1"] = 0 if okw.pop("separate_variant") else 1
    if "max_candidates" in okw:
        okw["max_iters"] = okw.pop("max_candidates")
    okw.pop("allow_truncation", None)
    ref = oracle.decode(g_obj, g_rot, g_scale, corner, res, pts, xyz, prob, cls,
                        oracle.DecodeParams.default(**okw))
    dg = t(cuda, g_obj).clone()
    hip = decode.decode_boxes(dg, t(cuda, g_rot), t(cuda, g_scale), t(cuda, pts), t(cuda, xyz),
                              t(cuda, prob), t(cuda, cls), res, corner=corner, mutate_grid=True, **kw)
    assert list(hip["cand_idx"]) == list(ref["cand_idx"])
    assert list(hip["verdict"]) == list(ref["verdict"])
    assert list(hip["classes"]) == list(ref["classes"])
    assert hip["boxes"].shape == ref["boxes"].shape
    np.testing.assert_allclose(hip["boxes"], ref["boxes"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hip["scores"], ref["scores"], rtol=0, atol=0)
    assert np.array_equal(dg.cpu().numpy(), ref["grid_obj_after"])      # same in-place suppression
    return ref, hip


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_decode_matches_oracle_on_oracle_grids(cuda, built_lib, seed):
    sc = make_scene(seed, n_points=8000)
    xyz, scale, prob, cls = synth_predictions(sc)
    pts = sc.points
    g = oracle.hv_forward(pts, xyz, scale, prob, sc.res, 120)
    corner, _, _ = oracle.grid_geometry(pts, sc.res)
    thr = float(np.sort(g[0].ravel())[-400])            # a few hundred cells above threshold
    ref, _ = compare(cuda, g[0], g[1], g[2], corner, sc.res, pts, xyz, prob, cls, thresh_high=thr)
    assert len(ref["cand_idx"]) >= 3
    compare(cuda, g[0], g[1], g[2], corner, sc.res, pts, xyz, prob, cls, thresh_high=thr,
            separate_variant=True)


def test_decode_planted_ties_and_rejections(cuda, built_lib):
    from tests.test_oracle_decode import box_points, planted
    res = 0.05
    rng = np.random.default_rng(0)
    g_obj, g_rot, g_scale = planted()
    A, B, C, D = (10, 10, 10), (10, 10, 30), (30, 10, 10), (30, 10, 30)
    g_obj[A] = 100; g_obj[B] = 100; g_obj[C] = 90; g_obj[D] = 80; g_obj[20, 4, 20] = 59.9
    g_rot[B] = [np.cos(0.7), np.sin(0.7)]
    half = np.array([0.2, 0.2, 0.2])
    pa, xa, qa = box_points(np.array(A) * res, half, 0.0, 200, rng)
    pb, xb, qb = box_points(np.array(B) * res, half, 0.7, 150, rng)
    pc, xc, qc = box_points(np.array(C) * res, half, 0.0, 100, rng, prob=0.1)
    pd, xd, qd = box_points(np.array(D) * res, half, 0.0, 100, rng)
    pts = np.concatenate([pa, pb, pc, pd]); xyz = np.concatenate([xa, xb, xc, -xd])
    prob = np.concatenate([qa, qb, qc, qd])
    cls = np.concatenate([np.full(200, 3), np.r_[np.full(70, 5), np.full(80, 2)], np.full(200, 1)]).astype(np.int32)
    ref, hip = compare(cuda, g_obj, g_rot, g_scale, np.zeros(3, np.float32), res, pts, xyz, prob, cls)
    assert list(hip["verdict"]) == [0, 0, 1, 2] and list(hip["classes"]) == [3, 2]


def test_decode_nothing_above_threshold_and_iteration_cap(cuda, built_lib):
    from tests.test_oracle_decode import planted
    g_obj, g_rot, g_scale = planted()
    pts = np.zeros((4, 3), np.float32)
    z = np.zeros(3, np.float32)
    ref, hip = compare(cuda, g_obj, g_rot, g_scale, z, 0.05, pts, pts, np.zeros(4, np.float32),
                       np.zeros(4, np.int32))
    assert len(hip["cand_idx"]) == 0 and len(hip["boxes"]) == 0
    g_scale[...] = 0.001
    g_obj[::6, ::6, ::6] = 77                          # many isolated peaks, cap at 5 candidates
    ref, hip = compare(cuda, g_obj, g_rot, g_scale, z, 0.05, pts, pts, np.zeros(4, np.float32),
                       np.zeros(4, np.int32), max_candidates=5, allow_truncation=True)
    assert len(hip["cand_idx"]) == 5 and hip["truncated"]
    # without allow_truncation the capped walk is never returned: it is redone with more room until the grid
    # maximum is below thresh_high, as the reference's `while True` does (eval_joint.py:204-209)
    ref = oracle.decode(g_obj, g_rot, g_scale, z, 0.05, pts, pts, np.zeros(4, np.float32), np.zeros(4, np.int32),
                        oracle.DecodeParams.default(max_iters=4096))
    full = decode.decode_boxes(t(cuda, g_obj), t(cuda, g_rot), t(cuda, g_scale), t(cuda, pts), t(cuda, pts),
                               t(cuda, np.zeros(4, np.float32)), t(cuda, np.zeros(4, np.int32)), 0.05, corner=z,
                               max_candidates=5)
    assert not full["truncated"] and len(full["cand_idx"]) == len(ref["cand_idx"]) > 5 * 8
    assert list(full["cand_idx"]) == list(ref["cand_idx"])


def test_detect_end_to_end_80k(cuda, built_lib):
    """vote (HIP) -> decode (HIP) -> NMS vs the oracle chain fed the HIP grids: box count, classes,
    candidate cells exact (BASELINE.json: bit-exact vote-grid indices / box counts)."""
    sc = make_scene(0, n_points=80000)
    xyz, scale, prob, cls = synth_predictions(sc)
    hv = HoughVoting(sc.res, 120)
    dets, raw = decode.detect(hv, t(cuda, sc.coords), t(cuda, xyz), t(cuda, scale), t(cuda, prob),
                              t(cuda, cls), sc.res)
    pts = sc.points
    with torch.no_grad():
        g = [o.cpu().numpy() for o in hv(t(cuda, pts), t(cuda, xyz), t(cuda, scale), t(cuda, prob))]
    corner, _, _ = oracle.grid_geometry(pts, sc.res)
    ref = oracle.decode(g[0], g[1], g[2], corner, sc.res, pts, xyz, prob, cls)
    # the two HIP votes may differ in the last bits (atomic order) but peaks are well separated
    assert len(raw["boxes"]) == len(ref["boxes"]) >= 6
    assert list(raw["classes"]) == list(ref["classes"])
    rd = oracle.nms_per_class(ref["boxes"], ref["scores"], ref["classes"])
    assert [d[0] for d in dets] == [d[0] for d in rd]
    # detections recover the planted boxes: every accepted box centre is near a ground-truth centre
    gt = sc.boxes[:, :3]
    for b in raw["boxes"]:
        assert np.min(np.linalg.norm(gt - b.mean(0)[None], axis=1)) < 0.15


@pytest.mark.parametrize("name", ["decode_ref_8k", "decode_ref_5k"])
def test_hip_path_matches_reference_lines_executed_on_cpu(cuda, built_lib, name):
    """head split + vote + decode on the GPU against the golden vectors the reference's own lines produced
    (eval_joint.py:173-190 and :195-263 exec()'d on CPU torch, tests/golden/make_decode_golden.py)"""
    import os
    from canonicalvoting_amd import pipeline
    from tests.golden.make_decode_golden import make_case
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    sc, F = make_case(int(z["seed"]), int(z["n"]), 1.0)
    res = float(z["res"])
    xyz, scale, prob, cls = pipeline.head_joint(t(cuda, F))
    assert np.array_equal(cls.cpu().numpy(), z["class_pred"].astype(np.int32))
    np.testing.assert_array_equal(xyz.cpu().numpy(), z["xyz_pred"])
    np.testing.assert_allclose(scale.cpu().numpy(), z["scale_pred"], rtol=2e-6)
    np.testing.assert_allclose(prob.cpu().numpy(), z["prob_pred"], rtol=2e-6, atol=1e-7)
    # vote + decode from the REFERENCE's head outputs (so a last-bit difference of expf cannot move a vote)
    th, tl, vr, el = z["consts"]
    pts = t(cuda, (sc.coords * np.float32(res)).astype(np.float32))
    hv = HoughVoting(res, 120)
    with torch.no_grad():
        g = hv(pts, t(cuda, z["xyz_pred"]), t(cuda, z["scale_pred"]), t(cuda, z["prob_pred"]))
    before = g[0].cpu().numpy().copy()
    raw = decode.decode_boxes(g[0], g[1], g[2], pts, t(cuda, z["xyz_pred"]), t(cuda, z["prob_pred"]),
                              t(cuda, z["class_pred"].astype(np.int32)), res, thresh_high=float(th), thresh_low=float(tl),
                              valid_ratio=float(vr), elimination=int(el), mutate_grid=True)
    assert len(raw["boxes"]) == len(z["boxes"]) and list(raw["classes"]) == list(z["classes"])
    np.testing.assert_allclose(raw["boxes"], z["boxes"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(raw["scores"], z["scores"], rtol=1e-6)
    zeroed = np.flatnonzero((before != 0) & (g[0].cpu().numpy() == 0))
    assert np.array_equal(zeroed, z["zeroed"].astype(zeroed.dtype))


def test_sorted_walker_of_the_big_grids_on_the_small_cases(cuda, built_lib):
    """Grids beyond 4 M cells (300k-point scenes) take dec_greedy_dispatch_big: the sorted register-resident walk
    (dec_greedy_sorted).  The oracle / planted-tie / iteration-cap / reference-line cases of this file are run through it
    too: CV_DEC_BIG_CELLS=1 sends every grid that way (read once per process, hence a fresh interpreter).  The 300k-point
    scene itself is compared with the oracle in test_production_size_gpu.py."""
    env = dict(os.environ, CV_DEC_BIG_CELLS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "oracle_grids or planted_ties or iteration_cap or reference_lines or end_to_end_80k"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
