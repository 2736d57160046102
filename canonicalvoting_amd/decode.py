"""Detection decode: the reference's inline loop as one function.

``decode_boxes`` is eval_joint.py:195-263 (greedy peak picking, grid suppression,
back-projection check, class vote); ``nms`` / ``get_iou_obb`` are eval_joint.py:75-89 and
utils/calc_map.py:6-21; ``detect`` chains vote -> decode -> per-class NMS the way
eval_joint.py:193-280 does.  Module-level constants keep the reference's names and values
(eval_joint.py:18-21).
"""
import ctypes

import numpy as np
import torch

from . import _lib, hv_cuda

thresh_high = 60
thresh_low = 10
valid_ratio = 0.2
elimination = 2


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def decode_boxes(grid_obj, grid_rot, grid_scale, scan_points, xyz_pred, prob_pred, class_pred, res,
                 corner=None, thresh_high=thresh_high, thresh_low=thresh_low,
                 valid_ratio=valid_ratio, elimination=elimination, prob_thresh=0.3, err_thresh=0.3,
                 separate_variant=False, max_candidates=512, mutate_grid=False, allow_truncation=False):
    """Greedy decode on the device, one host sync.

    scan_points [N,3] f32 world points (``curr_points * res``, eval_joint.py:200), ``corner``
    defaults to their minimum (``corners[0]``, :201).  Returns a dict with ``boxes`` [K,8,3],
    ``scores`` [K], ``classes`` [K] (numpy, acceptance order) plus the examined candidate cells
    and verdicts (0 accepted, 1 too few confident points :246-247, 2 LCC error :252-253).
    ``separate_variant`` selects eval_separate.py:209's elimination slice (no ``+1``).
    ``max_candidates`` is the first capacity of the walk, not a limit on the result: the reference's ``while True``
    (:204-209) runs until the grid maximum drops below ``thresh_high``, so a walk that fills its capacity with a cell
    >= thresh_high still live is redone with eight times the room (up to 65536, then RuntimeError).
    ``allow_truncation=True`` returns the capped walk instead, marked ``truncated=True`` in the dict."""
    L = _lib.lib()
    dev = grid_obj.device
    for t, name in ((grid_obj, "grid_obj"), (grid_rot, "grid_rot"), (grid_scale, "grid_scale"),
                    (scan_points, "scan_points"), (xyz_pred, "xyz_pred"), (prob_pred, "prob_pred")):
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32:
            raise RuntimeError("%s must be a contiguous float32 CUDA tensor" % name)
    cls = class_pred.to(torch.int32).contiguous()
    n = scan_points.shape[0]
    if corner is None:
        corner = hv_cuda.recent_corner(grid_obj)            # set by hv_cuda.forward (same points)
    if corner is None:
        corner, _, _ = hv_cuda.grid_geometry(scan_points, float(res))
    dims = (ctypes.c_int * 3)(*grid_obj.shape)
    p = _lib.DecodeParams(float(thresh_high), float(thresh_low), float(valid_ratio),
                          int(elimination), float(prob_thresh), 0 if separate_variant else 1,
                          int(max_candidates), float(err_thresh))
    M = int(max_candidates)
    while True:
        p.max_iters = M
        ws = _lib.scratch(dev, "decode", L.cv_decode_workspace_bytes(dims, n, M))
        n_cand, n_boxes, truncated = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        cand = np.zeros(M, np.int64)
        verdict = np.zeros(M, np.int32)
        boxes = np.zeros((M, 8, 3), np.float32)
        scores = np.zeros(M, np.float32)
        classes = np.zeros(M, np.int32)
        with torch.cuda.device(dev):
            rc = L.cv_decode_f32(
                _ptr(grid_obj), _ptr(grid_rot), _ptr(grid_scale), dims,
                (ctypes.c_float * 3)(*[float(v) for v in corner]), ctypes.c_float(float(res)),
                _ptr(scan_points), _ptr(xyz_pred), _ptr(prob_pred), _ptr(cls), n, ctypes.byref(p),
                1 if mutate_grid else 0, _ptr(ws), ws.numel(), ctypes.byref(n_cand),
                cand.ctypes.data_as(_lib.c_i64_p), verdict.ctypes.data_as(_lib.c_i32_p),
                ctypes.byref(n_boxes), boxes.ctypes.data_as(_lib.c_float_p),
                scores.ctypes.data_as(_lib.c_float_p), classes.ctypes.data_as(_lib.c_i32_p),
                ctypes.byref(truncated), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "cv_decode_f32")
        if not truncated.value or allow_truncation:
            break
        # the reference's `while True` (eval_joint.py:204-209) runs until the grid maximum drops below thresh_high:
        # a capped walk that stopped early is redone with more room, never returned as if it were complete
        if M >= 65536 or mutate_grid:
            raise RuntimeError("decode_boxes: more than %d candidate cells >= thresh_high (max_candidates)" % M)
        M = min(M * 8, 65536)
    k, m = n_boxes.value, n_cand.value
    return dict(boxes=boxes[:k].copy(), scores=scores[:k].copy(), classes=classes[:k].copy(),
                cand_idx=cand[:m].copy(), verdict=verdict[:m].copy(), truncated=bool(truncated.value))


def get_iou_obb(bbox1, bbox2):
    """utils/calc_map.py:6-21 on two [8,3] corner arrays."""
    a = np.ascontiguousarray(bbox1, np.float32)
    b = np.ascontiguousarray(bbox2, np.float32)
    return float(_lib.lib().cv_iou_obb(a.ctypes.data_as(_lib.c_float_p), b.ctypes.data_as(_lib.c_float_p)))


def nms(boxes, scores, overlap_threshold):
    """eval_joint.py:75-89: indices kept, highest score first."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8, 3)
    s = np.ascontiguousarray(scores, np.float32)
    n = int(s.shape[0])
    pick = np.zeros(max(n, 1), np.int32)
    k = _lib.lib().cv_nms_obb(b.ctypes.data_as(_lib.c_float_p), s.ctypes.data_as(_lib.c_float_p), n,
                              float(overlap_threshold), pick.ctypes.data_as(_lib.c_i32_p))
    if k < 0:
        _lib.check(k, "cv_nms_obb")
    return [int(v) for v in pick[:k]]


def nms_per_class(boxes, scores, classes, nclasses=9, overlap_threshold=0.3):
    """eval_joint.py:270-280 -> [(class, box[8,3], score)] in the reference's order."""
    out = []
    if len(classes) == 0:
        return out
    for i in range(nclasses):
        sel = classes == i
        if sel.sum() > 0:
            bc, sc = boxes[sel], scores[sel]
            for j in nms(bc, sc, overlap_threshold):
                out.append((i, bc[j], float(sc[j])))
    return out


def detect(hv, coords_int, xyz_pred, scale_pred, prob_pred, class_pred, res, nclasses=9, scan_points=None, **kw):
    """eval_joint.py:193-280 after the network: vote, decode, per-class NMS.

    ``hv`` is a HoughVoting module, ``coords_int`` the [N,3] integer voxel coordinates
    (``scan_points[:, 1:]``).  Returns (detections, raw) where detections is the
    ``map_scene`` list of (class, box, score)."""
    if scan_points is None:
        scan_points = (coords_int.to(xyz_pred.device) * res).float().contiguous()    # :193,:200
    with torch.no_grad():
        grid_obj, grid_rot, grid_scale = hv(scan_points, xyz_pred.contiguous(),
                                            scale_pred.contiguous(), prob_pred.contiguous())
    raw = decode_boxes(grid_obj, grid_rot, grid_scale, scan_points, xyz_pred.contiguous(),
                       prob_pred.contiguous(), class_pred, res, **kw)
    dets = nms_per_class(raw["boxes"], raw["scores"], raw["classes"], nclasses)
    return dets, raw
