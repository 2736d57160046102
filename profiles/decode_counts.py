"""list length (cells >= thresh_high) and candidates examined by the decode of the bench scenes (teacher-fed)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import decode
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for large in (False, True):
    sc = make_scene(0, n_points=300000, room=(9.0, 3.0, 9.0), n_boxes=40) if large else make_scene(0, n_points=80000)
    xyz, scale, prob, cls = synth_predictions(sc)
    hv = HoughVoting(sc.res, 120)
    with torch.no_grad():
        g = hv(t(sc.points), t(xyz), t(scale), t(prob))
    th = float(os.environ.get("TH", str(decode.thresh_high)))
    n_list = int((g[0] >= th).sum())
    dets, raw = decode.detect(hv, t(sc.coords), t(xyz), t(scale), t(prob), t(cls), sc.res, thresh_high=th)
    print("300k" if large else "80k", "cells >= thresh_high:", n_list, "candidates examined:", len(raw["cand_idx"]), "boxes:", len(raw["boxes"]), "grid", tuple(g[0].shape))
