#!/usr/bin/env python
"""eval_joint.py counterpart (reference eval_joint.py:137-312) on synthetic scans: network -> head -> vote ->
decode -> NMS per scene, then mAP @0.25 / @0.5.  argparse instead of hydra (absent here).

    python scripts/eval_joint.py [--scenes 4] [--points 80000] [--weights joint.pth] [--teacher]

--teacher feeds the vote/decode stage with predictions synthesised from the labels (there is no trained
checkpoint offline); the network forward still runs.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import calc_map, decode, pipeline  # noqa: E402
from canonicalvoting_amd import me as ME  # noqa: E402
from canonicalvoting_amd.data import (ScanNetXYZProbMultiDataset, SyntheticScanDataset, collate_fn,  # noqa: E402
                                     load_config)
from canonicalvoting_amd.hough import HoughVoting  # noqa: E402
from canonicalvoting_amd.minkunet import MinkUNet34C  # noqa: E402
from canonicalvoting_amd.synth import synth_predictions  # noqa: E402


def evaluate(model, dataset, res=0.03, teacher=False, nclasses=9, device="cuda"):
    hv = HoughVoting(res)
    pred_map_cls, gt_map_cls = {}, {}
    loader = torch.utils.data.DataLoader(dataset, collate_fn=collate_fn, batch_size=1, shuffle=False)
    for index, (ids, coords, feats, _, _, _) in enumerate(loader):
        id_scan = ids[0]
        feats = feats.to(device)
        feats[:, -3:] = feats[:, -3:] * 2.0 - 1.0                             # eval_joint.py:167-168: colour columns only
        coords = coords.to(device)
        with torch.no_grad():
            out = model(ME.SparseTensor(feats, coords, device=device))
            xyz, scale, prob, cls = pipeline.head_joint(out.F, nclasses)
        if teacher:
            t = lambda a: torch.from_numpy(a).to(device)
            xyz, scale, prob, cls = [t(a) for a in synth_predictions(dataset.scene(index))]
        dets, _ = decode.detect(hv, coords[:, 1:], xyz, scale, prob, cls, res, nclasses)
        pred_map_cls[id_scan] = dets
        gt_map_cls[id_scan] = [(c, calc_map.gt_box(*p)) for c, p in dataset.gt(index)]     # eval_joint.py:285-301
    return {thr: calc_map.compute_map(pred_map_cls, gt_map_cls, thr) for thr in (0.25, 0.5)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--points", type=int, default=80000)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--teacher", action="store_true")
    ap.add_argument("--config", default=None, help="the reference's config.yaml: evaluate on real ScanNet/Scan2CAD files")
    a = ap.parse_args()
    cfg = load_config(a.config, category="all") if a.config else None
    model = MinkUNet34C(6 if (cfg and cfg.use_xyz) else 3, 6 * 9 + 9 + 1)
    if a.weights:
        model.load_state_dict(torch.load(a.weights, map_location="cpu"))
    model = model.cuda().eval()
    if cfg:
        res = evaluate(model, ScanNetXYZProbMultiDataset(cfg, training=False, augment=False), res=cfg.scannet_res)
    else:
        res = evaluate(model, SyntheticScanDataset(a.scenes, a.points, seed0=100), teacher=a.teacher)
    for thr, r in res.items():
        print("IoU %.2f: mAP %.4f  AR %.4f" % (thr, r["mAP"], r["AR"]))


if __name__ == "__main__":
    main()
