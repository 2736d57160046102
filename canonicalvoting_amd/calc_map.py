"""Detection mAP evaluator: the step AFTER the hot path.

Restates utils/calc_map.py:40-226 (VOC AP, per-class precision/recall, class fan-out) and
eval_joint.py:92-109 (``compute_map``) / :285-301 (ground-truth box from ``tx ty tz ry sx sy sz``) in plain
numpy, single process (the reference's ``Pool(10)`` only parallelises classes).  The IoU is
``decode.get_iou_obb`` (utils/calc_map.py:6-21 without shapely).

Deviation: the reference indexes its per-class results with ``enumerate(gt.keys())`` although only the
classes that have predictions were evaluated (utils/calc_map.py:214-219), which mis-assigns results once a
ground-truth class has no prediction; here every class gets its own result.
"""
import numpy as np

BBOX_RAW = np.array([[1, 1, 1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1], [-1, -1, -1],
                     [-1, -1, 1]], np.float32)          # eval_joint.py:202-203 corner order


def voc_ap(rec, prec, use_07_metric=False):
    """utils/calc_map.py:40-71"""
    rec, prec = np.asarray(rec, float), np.asarray(prec, float)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            sel = rec >= t
            ap += (prec[sel].max() if sel.any() else 0.0) / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]              # precision envelope
    step = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1]))


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, get_iou_func=None):
    """utils/calc_map.py:78-168.  pred {img: [(box, score)]}, gt {img: [box]} -> rec, prec, ap"""
    if get_iou_func is None:
        from .decode import get_iou_obb as get_iou_func
    recs = {img: {"bbox": np.array(boxes), "det": [False] * len(boxes)} for img, boxes in gt.items()}
    npos = sum(len(b) for b in gt.values())
    for img in pred:
        recs.setdefault(img, {"bbox": np.array([]), "det": []})
    ids, conf, boxes = [], [], []
    for img, dets in pred.items():
        for box, score in dets:
            ids.append(img); conf.append(score); boxes.append(box)
    conf = np.array(conf)
    order = np.argsort(-conf)
    nd = len(ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, k in enumerate(order):
        R = recs[ids[k]]
        bb = np.asarray(boxes[k], float)
        ovmax, jmax = -np.inf, -1
        G = R["bbox"].astype(float)
        if G.size > 0:
            for j in range(G.shape[0]):
                iou = get_iou_func(bb, G[j])
                if iou > ovmax:
                    ovmax, jmax = iou, j
        if ovmax > ovthresh and not R["det"][jmax]:
            tp[d] = 1.0
            R["det"][jmax] = True
        else:
            fp[d] = 1.0
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None):
    """utils/calc_map.py:177-226 without the process pool.
    pred_all {img: [(cls, box, score)]}, gt_all {img: [(cls, box)]} -> rec, prec, ap dicts by class"""
    pred, gt = {}, {}
    for img, dets in pred_all.items():
        for cls, box, score in dets:
            pred.setdefault(cls, {}).setdefault(img, []).append((box, score))
            gt.setdefault(cls, {}).setdefault(img, [])
    for img, objs in gt_all.items():
        for cls, box in objs:
            gt.setdefault(cls, {}).setdefault(img, []).append(box)
    rec, prec, ap = {}, {}, {}
    for cls in gt:
        if cls in pred:
            rec[cls], prec[cls], ap[cls] = eval_det_cls(pred[cls], gt[cls], ovthresh, use_07_metric, get_iou_func)
        else:
            rec[cls], prec[cls], ap[cls] = 0, 0, 0
    return rec, prec, ap


def compute_map(pred_map_cls, gt_map_cls, ovthresh=0.5, get_iou_func=None):
    """eval_joint.py:92-109"""
    rec, prec, ap = eval_det(pred_map_cls, gt_map_cls, ovthresh=ovthresh, get_iou_func=get_iou_func)
    ret = {}
    for key in sorted(ap.keys(), key=str):
        ret["%s Average Precision" % str(key)] = ap[key]
    ret["mAP"] = float(np.mean(list(ap.values()))) if ap else 0.0
    recalls = []
    for key in sorted(ap.keys(), key=str):
        r = rec[key][-1] if hasattr(rec[key], "__len__") and len(rec[key]) else 0
        ret["%s Recall" % str(key)] = r
        recalls.append(r)
    ret["AR"] = float(np.mean(recalls)) if recalls else 0.0
    return ret


def gt_box(tx, ty, tz, ry, sx, sy, sz):
    """eval_joint.py:287-297: corners of a ground-truth box (half extents sx,sy,sz, yaw ry about y)."""
    c, s = np.cos(ry), np.sin(ry)
    R = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
    return (R @ np.diag([sx, sy, sz]) @ BBOX_RAW.astype(np.float64).T).T + np.array([tx, ty, tz])
