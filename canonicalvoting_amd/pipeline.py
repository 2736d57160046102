"""The eval_joint.py per-scene path as functions: network -> head split -> vote -> decode -> NMS
(eval_joint.py:163-280), everything on the device, three host syncs per scene (coordinate-set
sizes, vote-grid shape, decode results)."""
import ctypes

import torch

from . import _lib, decode, hv_cuda
from . import me as ME


def head_joint(out_feats, nclasses=9, log_scale=True):
    """eval_joint.py:173-190 in one kernel: -> xyz[N,3], scale[N,3], prob[N], class[N] (int32)."""
    L = _lib.lib()
    F = out_feats.contiguous()
    n, dev = F.shape[0], F.device
    xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((n, 3), dtype=torch.float32, device=dev)
    prob = torch.empty((n,), dtype=torch.float32, device=dev)
    cls = torch.empty((n,), dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(L.cv_head_joint_f32(p(F), n, F.stride(0), nclasses, 1 if log_scale else 0, p(xyz), p(scale),
                                       p(prob), p(cls), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "cv_head_joint_f32")
    return xyz, scale, prob, cls


def detect_scene(model, hv, coords4, feats, res, nclasses=9, log_scale=True, **decode_kw):
    """coords4 [N,4] int (batch 0), feats [N,C] already recentred (eval_joint.py:167-168).
    Returns (detections [(class, box[8,3], score)], raw decode dict, network output)."""
    with torch.no_grad():
        # eval_joint.py:193 scan points; their bounds reduction (the vote grid shape) runs under the network
        scan_points = (coords4[:, 1:].to(feats.device) * res).float().contiguous()
        hv_cuda.prefetch_geometry(scan_points)
        x = ME.SparseTensor(feats, coords4, device=feats.device)
        deferred = hasattr(model, "check_range")
        y = model(x, defer_check=True) if deferred else model(x)
        for attempt in range(2):
            xyz, scale, prob, cls = head_joint(y.F, nclasses, log_scale)
            dets, raw = decode.detect(hv, coords4[:, 1:], xyz, scale, prob, cls, res, nclasses,
                                      scan_points=scan_points, **decode_kw)
            # decode has waited for the stream: the fp16-range flag of the network's convolutions is final now
            y2 = model.check_range(x, y) if (deferred and attempt == 0) else y
            if y2 is y:
                break
            y = y2
    return dets, raw, y


def head_separate(out_feats, log_scale=True):
    """eval_separate.py:170-181 in one kernel: -> xyz[N,3], scale[N,3], prob[N]."""
    L = _lib.lib()
    F = out_feats.contiguous()
    n, dev = F.shape[0], F.device
    xyz = torch.empty((n, 3), dtype=torch.float32, device=dev)
    scale = torch.empty((n, 3), dtype=torch.float32, device=dev)
    prob = torch.empty((n,), dtype=torch.float32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(L.cv_head_separate_f32(p(F), n, F.stride(0), 1 if log_scale else 0, p(xyz), p(scale), p(prob),
                                          ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "cv_head_separate_f32")
    return xyz, scale, prob


def detect_scene_separate(models, hv, coords4, feats, res, log_scale=True, overlap_threshold=0.3, **decode_kw):
    """eval_separate.py:162-264: one 8-channel model per category on the SAME SparseTensor (the
    coordinate sets, kernel maps and mask orders are built once and shared by all models - the
    reference rebuilds nothing either, but here it is one hash/map build per scene, not per model),
    vote + decode (eval_separate.py:209 elimination slice, float32 error threshold) and NMS per category.
    models: dict category -> MinkUNet34C(in, 8).  Returns [(category, box[8,3], score)]."""
    import numpy as np
    decode_kw.setdefault("separate_variant", True)
    decode_kw.setdefault("err_thresh", float(np.float32(0.3)))     # tensor > 0.3 compares in float32 (:252)
    out = []
    with torch.no_grad():
        x = ME.SparseTensor(feats, coords4, device=feats.device)
        scan_points = (coords4[:, 1:].to(feats.device) * res).float().contiguous()
        zeros_cls = torch.zeros(coords4.shape[0], dtype=torch.int32, device=feats.device)
        for category, model in models.items():
            xyz, scale, prob = head_separate(model(x).F, log_scale)
            grid_obj, grid_rot, grid_scale = hv(scan_points, xyz, scale, prob)
            raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, scan_points, xyz, prob, zeros_cls, res,
                                      **decode_kw)
            for i in decode.nms(raw["boxes"], raw["scores"], overlap_threshold):
                out.append((category, raw["boxes"][i], float(raw["scores"][i])))
    return out
