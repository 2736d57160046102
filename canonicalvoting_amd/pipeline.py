"""The eval_joint.py per-scene path as functions: network -> head split -> vote -> decode -> NMS
(eval_joint.py:163-280), everything on the device, three host syncs per scene (coordinate-set
sizes, vote-grid shape, decode results)."""
import ctypes

import torch

from . import _lib, decode
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
        x = ME.SparseTensor(feats, coords4, device=feats.device)
        y = model(x)
        xyz, scale, prob, cls = head_joint(y.F, nclasses, log_scale)
        dets, raw = decode.detect(hv, coords4[:, 1:], xyz, scale, prob, cls, res, nclasses, **decode_kw)
    return dets, raw, y
