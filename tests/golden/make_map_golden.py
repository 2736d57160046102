"""Golden vectors for the mAP evaluator, produced BY THE REFERENCE'S OWN CODE run here:
utils/calc_map.py (pure numpy apart from shapely) is imported from /root/reference with a stub
`shapely.geometry` module, and its voc_ap / eval_det_cls are run on seeded random detections with this
repo's OBB IoU passed in as get_iou_func.  Only inputs and outputs are stored (tests/golden/map_golden.npz).

    python tests/golden/make_map_golden.py        # needs /root/reference (build container only)
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
shapely = types.ModuleType("shapely"); geometry = types.ModuleType("shapely.geometry")
geometry.Polygon = object
shapely.geometry = geometry
sys.modules["shapely"] = shapely; sys.modules["shapely.geometry"] = geometry
sys.path.insert(0, "/root/reference")
from utils import calc_map as ref          # noqa: E402  (the reference's evaluator)

from canonicalvoting_amd import calc_map as mine   # noqa: E402
from canonicalvoting_amd.decode import get_iou_obb  # noqa: E402

rng = np.random.default_rng(7)
out = {}
# voc_ap known answers (0.35 and 11-point 0.40909)
out["ap_a"] = ref.voc_ap(np.array([.1, .2, .2, .4]), np.array([1, 1, .66, .75]))
out["ap_a07"] = ref.voc_ap(np.array([.1, .2, .2, .4]), np.array([1, 1, .66, .75]), True)
recs, precs, aps, aps07 = [], [], [], []
for k in range(6):
    n = int(rng.integers(3, 30))
    rec = np.sort(rng.uniform(0, 1, n)); prec = rng.uniform(0, 1, n)
    recs.append(rec); precs.append(prec); aps.append(ref.voc_ap(rec, prec)); aps07.append(ref.voc_ap(rec, prec, True))
out["rand_rec"] = np.array(recs, dtype=object); out["rand_prec"] = np.array(precs, dtype=object)
out["rand_ap"] = np.array(aps); out["rand_ap07"] = np.array(aps07)


def scene(nimg, ngt, npred):
    gt, pred = {}, {}
    for i in range(nimg):
        g = [mine.gt_box(*rng.uniform(0, 4, 3), rng.uniform(0, 6.28), *rng.uniform(0.2, 0.7, 3)) for _ in range(ngt)]
        gt["s%d" % i] = g
        p = []
        for b in g[:max(1, ngt - 1)]:
            p.append((b + rng.normal(0, 0.05, 3), float(rng.uniform(0.3, 1))))
        for _ in range(npred):
            p.append((mine.gt_box(*rng.uniform(0, 4, 3), rng.uniform(0, 6.28), *rng.uniform(0.2, 0.7, 3)),
                      float(rng.uniform(0, 1))))
        pred["s%d" % i] = p
    return pred, gt


cases = []
for k, (nimg, ngt, npred) in enumerate(((3, 4, 3), (5, 2, 6), (2, 6, 1))):
    pred, gt = scene(nimg, ngt, npred)
    for thr in (0.25, 0.5):
        r, p, ap = ref.eval_det_cls({k2: list(v) for k2, v in pred.items()}, {k2: list(v) for k2, v in gt.items()},
                                    thr, False, get_iou_obb)
        cases.append(dict(pred=pred, gt=gt, thr=thr, rec=r, prec=p, ap=ap))
out["cases"] = np.array(cases, dtype=object)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "map_golden.npz"), **out)
print("voc_ap", out["ap_a"], out["ap_a07"], "cases", len(cases), [round(c["ap"], 4) for c in cases])
