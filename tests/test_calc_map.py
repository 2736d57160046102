"""mAP evaluator vs golden vectors produced by the reference's own utils/calc_map.py
(tests/golden/make_map_golden.py imported it here with shapely stubbed)."""
import os

import numpy as np

from canonicalvoting_amd import calc_map

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "map_golden.npz"), allow_pickle=True)


def test_voc_ap_known_answers_from_reference():
    rec, prec = np.array([.1, .2, .2, .4]), np.array([1, 1, .66, .75])
    assert abs(calc_map.voc_ap(rec, prec) - float(Z["ap_a"])) < 1e-12 and abs(float(Z["ap_a"]) - 0.35) < 1e-12
    assert abs(calc_map.voc_ap(rec, prec, True) - float(Z["ap_a07"])) < 1e-12
    for r, p, a, a07 in zip(Z["rand_rec"], Z["rand_prec"], Z["rand_ap"], Z["rand_ap07"]):
        assert abs(calc_map.voc_ap(r, p) - a) < 1e-12
        assert abs(calc_map.voc_ap(r, p, True) - a07) < 1e-12


def test_eval_det_cls_matches_reference(built_lib):
    for c in Z["cases"]:
        rec, prec, ap = calc_map.eval_det_cls(c["pred"], c["gt"], c["thr"])
        np.testing.assert_allclose(rec, c["rec"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(prec, c["prec"], rtol=0, atol=1e-12)
        assert abs(ap - c["ap"]) < 1e-12


def test_compute_map_end_to_end(built_lib):
    b = calc_map.gt_box(1, 0.5, 1, 0.3, 0.4, 0.5, 0.3)
    far = calc_map.gt_box(5, 0.5, 5, 0.0, 0.4, 0.5, 0.3)
    pred = {"a": [("chair", b, 0.9), ("chair", far, 0.8), ("table", far, 0.5)]}
    gt = {"a": [("chair", b), ("table", b), ("sofa", b)]}
    r = calc_map.compute_map(pred, gt, 0.5)
    assert r["chair Average Precision"] == 1.0 and r["table Average Precision"] == 0.0
    assert r["sofa Average Precision"] == 0 and abs(r["mAP"] - 1 / 3) < 1e-12
    # box convention: rows 0-3 top face, row 4 bottom (utils/calc_map.py:13-18)
    assert b[0, 1] > b[4, 1] and np.allclose(b.mean(0), [1, 0.5, 1])
