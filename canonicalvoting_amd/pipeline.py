"""The eval_joint.py per-scene path as functions: network -> head split -> vote -> decode -> NMS
(eval_joint.py:163-280), everything on the device, three host syncs per scene (coordinate-set
sizes, vote-grid shape, decode results)."""
import collections
import contextlib
import ctypes
import os

import torch

from . import _lib, decode, hv_cuda
from . import me as ME


class ScenePolicy(collections.namedtuple("ScenePolicy", "conv_split_target vote_part_records masked_min_rows")):
    """Launch sizing of ONE scene call - the three choices that depend on how many scenes the host keeps in flight:
    conv_split_target  workgroups a split coarse-level convolution aims at (0: the library's value, 768)
    vote_part_records  records one workgroup of a hot (tile, plane) of the vote takes (0: the library's value, 4096)
    masked_min_rows    rows from which a level's 3x3x3 convolutions run mask-sorted (library: 16384)
    Every integer output is the same under every policy; the network output moves in fp32 summation order only
    (tests/test_production_size_gpu.py runs the parity cases under both policies).  A policy travels WITH the call -
    cv_scene_desc.conv_split_target / vote_part_records / masked_min_rows for detect_scene_c, the calling thread's values inside
    scene_policy(...) for the call-by-call path - so two hosts with different policies in one process do not see each other."""
    __slots__ = ()

    def as_config(self):
        return {"conv_split_target": self.conv_split_target, "vote_part_records": self.vote_part_records,
                "masked_min_rows": self.masked_min_rows}


def policy_for_scenes_in_flight(n):
    """The launch sizing for a host that keeps `n` scenes in flight on separate streams (one thread + stream per scene, as
    bench.py does).  The library's defaults are the best for ONE scene at a time; from four scenes in flight the other scenes
    fill the chip and three choices turn (measured on MI355X, LABNOTES rounds 3 and 5): the split convolutions aim at 256
    workgroups instead of 768, a hot (tile, plane) of the vote takes 12288 records per workgroup instead of 4096, and the
    3x3x3 convolutions run mask-sorted from 8192 rows instead of 16384."""
    lib_rows = ME.CoordinateManager.LIB_MASKED_MIN_ROWS
    if int(n) >= 4:
        return ScenePolicy(256, 12288, min(8192, lib_rows))
    return ScenePolicy(0, 0, lib_rows)


@contextlib.contextmanager
def scene_policy(policy):
    """The calling thread's launches inside the block run under `policy` (None: no change): what detect_scene and the module
    paths read (cv_sp_set_split_target_thread, cv_hv_set_part_records_thread, ME.masked_min_rows()); restored on exit."""
    if policy is None:
        yield
        return
    L = _lib.lib()
    prev_t = L.cv_sp_set_split_target_thread(int(policy.conv_split_target))
    prev_r = L.cv_hv_set_part_records_thread(int(policy.vote_part_records))
    prev_m = ME.set_masked_min_rows_thread(policy.masked_min_rows)
    try:
        yield
    finally:
        L.cv_sp_set_split_target_thread(prev_t)
        L.cv_hv_set_part_records_thread(prev_r)
        ME.set_masked_min_rows_thread(prev_m)


def configure_for_scenes_in_flight(n, model=None):
    """policy_for_scenes_in_flight(n) installed PROCESS-WIDE (the default of every thread that runs without a policy of its
    own): for a process with one host loop.  Call it before the scene threads start; hosts that share a process pass
    `policy=` per call instead.  `model` is not needed any more (the models follow ME.masked_min_rows()).  Returns the
    settings as a dict."""
    pol = policy_for_scenes_in_flight(n)
    ME.set_split_target(pol.conv_split_target)
    _lib.lib().cv_hv_set_part_records(pol.vote_part_records)
    ME.CoordinateManager.MASKED_MIN_ROWS = pol.masked_min_rows
    return pol.as_config()


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


def detect_scene(model, hv, coords4, feats, res, nclasses=9, log_scale=True, policy=None, **decode_kw):
    """coords4 [N,4] int (batch 0), feats [N,C] already recentred (eval_joint.py:167-168).  policy: ScenePolicy of this call.
    Returns (detections [(class, box[8,3], score)], raw decode dict, network output)."""
    with torch.no_grad(), scene_policy(policy):
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


class _SceneHost:
    """per-(device, stream) host-side scratch of detect_scene_c: pinned landing words and the result arrays"""

    def __init__(self, max_candidates):
        import numpy as np
        self.M = max_candidates
        self.pinned = torch.empty(256, dtype=torch.uint8).pin_memory()
        self.cand = np.zeros(max_candidates, np.int64)
        self.verdict = np.zeros(max_candidates, np.int32)
        self.boxes = np.zeros((max_candidates, 8, 3), np.float32)
        self.scores = np.zeros(max_candidates, np.float32)
        self.classes = np.zeros(max_candidates, np.int32)
        self.pick = np.zeros(max_candidates, np.int32)
        self.ws_hint = 0
        self.grid_hint = 0
        self.last_host_us = (0.0, 0.0, 0.0, 0.0)


_scene_hosts = {}
_scene_lock = __import__("threading").Lock()


def detect_scene_c(model, hv, coords4, feats, res, nclasses=9, log_scale=True, scan_points=None, predictions=None,
                   max_candidates=512, keep=None, events=None, adaptive_split=False, policy=None, **decode_kw):
    """detect_scene through ONE C call (cv_detect_scene_f32: coordinate plan -> network program -> head -> vote -> decode
    -> per-class NMS; two host waits inside it, the GIL released for its whole duration).  Same kernels in the same order
    as detect_scene: bit-identical results (tests/test_scene_call_gpu.py).  ``predictions`` = (xyz, scale, prob, class)
    fed to vote + decode instead of the network's (bench.py --predictions teacher).  A scene that needs what the call
    does not do - a range fallback onto the bf16 triples, a decode walk beyond ``max_candidates`` - is redone by the
    call-by-call path.  ``events``: five recorded torch.cuda.Event that the call re-records at the stage boundaries.
    ``policy``: ScenePolicy of THIS call (None: the thread's / process-wide values).
    Returns (detections, raw decode dict, network output [N, C])."""
    import numpy as np
    L = _lib.lib()
    dev = feats.device
    n = coords4.shape[0]
    if scan_points is None:
        scan_points = (coords4[:, 1:].to(dev) * res).float().contiguous()
    pieces = 1 if ME.COMPUTE_DTYPE == "bf16" else model.PIECES
    c_ops, c_bufs, _ = model._program(dev, pieces)
    cm_cls = ME.CoordinateManager
    G = cm_cls.MASK_GROUPS if (27 + cm_cls.MASK_GROUPS - 1) // cm_cls.MASK_GROUPS <= 10 else 0
    coords4 = coords4.to(device=dev, dtype=torch.int32).contiguous()
    feats = feats.contiguous()
    y = torch.empty((n, model.final.out_channels), dtype=torch.float32, device=dev)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    with _scene_lock:
        host = _scene_hosts.get(key)
        if host is None or host.M < max_candidates:
            host = _scene_hosts[key] = _SceneHost(max_candidates)
    d = _lib.SceneDesc()
    vp = ctypes.c_void_p
    d.d_coords4, d.n, d.d_feats, d.feats_ld = vp(coords4.data_ptr()), n, vp(feats.data_ptr()), feats.stride(0)
    d.d_points, d.res, d.num_rots = vp(scan_points.data_ptr()), float(res), hv_cuda._scalar(hv.num_rots, "i")
    d.ops, d.n_ops, d.bufs, d.n_bufs = ctypes.cast(c_ops, vp), len(c_ops), ctypes.cast(c_bufs, vp), len(c_bufs)
    d.stem_k, d.mask_groups = model.conv0p1s1.kernel_size, G
    d.masked_min_rows = policy.masked_min_rows if policy is not None else model.masked_min_rows()
    if policy is not None:
        d.conv_split_target, d.vote_part_records = int(policy.conv_split_target), int(policy.vote_part_records)
    d.max_channels, d.use_range_flag = max(model.PLANES), 1 if pieces == 2 else 0
    d.d_out_feats, d.out_ld, d.out_channels = vp(y.data_ptr()), y.stride(0), y.shape[1]
    d.nclasses, d.log_scale = nclasses, 1 if log_scale else 0
    if predictions is not None:
        px, ps, pp, pc = predictions
        pc = pc.to(torch.int32).contiguous()
        d.d_xyz_in, d.d_scale_in, d.d_prob_in, d.d_class_in = (vp(px.data_ptr()), vp(ps.data_ptr()), vp(pp.data_ptr()),
                                                                 vp(pc.data_ptr()))
    d.vote_algo = hv_cuda._algo
    p = d.decode
    p.thresh_high = float(decode_kw.get("thresh_high", decode.thresh_high))
    p.thresh_low = float(decode_kw.get("thresh_low", decode.thresh_low))
    p.valid_ratio = float(decode_kw.get("valid_ratio", decode.valid_ratio))
    p.elimination = int(decode_kw.get("elimination", decode.elimination))
    p.prob_thresh = float(decode_kw.get("prob_thresh", 0.3))
    p.elim_hi_plus1 = 0 if decode_kw.get("separate_variant", False) else 1
    p.err_thresh = float(decode_kw.get("err_thresh", 0.3))
    d.max_candidates, d.nms_threshold = host.M, 0.3
    d.adaptive_split = 1 if adaptive_split else 0
    if events is not None:          # five torch.cuda.Event (recorded once, so that their handles exist): scene start, after
        for i in range(5):          # the network, after the head split, after the vote, after the decode
            d.events[i] = events[i].cuda_event
    d.h_pinned, d.pinned_bytes = vp(host.pinned.data_ptr()), host.pinned.numel()
    d.h_cand_idx, d.h_verdict = vp(host.cand.ctypes.data), vp(host.verdict.ctypes.data)
    d.h_boxes, d.h_scores, d.h_classes, d.h_pick = (vp(host.boxes.ctypes.data), vp(host.scores.ctypes.data),
                                                    vp(host.classes.ctypes.data), vp(host.pick.ctypes.data))
    r = _lib.SceneResult()
    grids_t = None
    if os.environ.get("CV_SCENE_GRIDS", "ws") == "torch" and host.grid_hint:
        grids_t = torch.empty(host.grid_hint, dtype=torch.float32, device=dev)      # experiment: the grids as a fresh allocation
        d.d_grids, d.grid_capacity_floats = vp(grids_t.data_ptr()), grids_t.numel()
    need = max(host.ws_hint, 64 << 20)
    for attempt in range(4):
        ws = _lib.scratch(dev, "scene_call", need)
        d.d_ws, d.ws_bytes = vp(ws.data_ptr()), ws.numel()
        with torch.cuda.device(dev):
            rc = L.cv_detect_scene_f32(ctypes.byref(d), ctypes.byref(r), vp(torch.cuda.current_stream(dev).cuda_stream))
        if rc == -12 and r.needed_ws_bytes > ws.numel():            # CV_ENOMEM: grow the scratch and run the scene again
            torch.cuda.current_stream(dev).synchronize()
            need = int(r.needed_ws_bytes)
            continue
        _lib.check(rc, "cv_detect_scene_f32")
        break
    else:
        _lib.check(rc, "cv_detect_scene_f32")          # still CV_ENOMEM after four growing attempts: not an empty scene
    host.ws_hint = max(host.ws_hint, int(r.needed_ws_bytes))
    host.last_host_us = tuple(r.host_us)            # where the call's host time went (plan + wait, network enqueue, head + vote, decode + wait)
    host.grid_hint = max(host.grid_hint, int(r.needed_grid_floats) * 9 // 8)
    if r.range_flag or r.truncated:
        # rare: a convolution input beyond the fp16 range, or more candidate cells than the result arrays hold
        if r.range_flag:
            model.range_fallbacks = getattr(model, "range_fallbacks", 0) + 1
        with torch.no_grad(), scene_policy(policy):
            x = ME.SparseTensor(feats, coords4, device=dev)
            yy = model.program_forward(x, pieces=3) if r.range_flag else x._like(y, 1)
            pred = head_joint(yy.F, nclasses, log_scale) if predictions is None else predictions
            dets, raw = decode.detect(hv, coords4[:, 1:], pred[0], pred[1], pred[2], pred[3], res, nclasses,
                                      scan_points=scan_points, **decode_kw)
        return dets, raw, yy.F
    k, m = r.n_boxes, r.n_cand
    raw = dict(boxes=host.boxes[:k].copy(), scores=host.scores[:k].copy(), classes=host.classes[:k].copy(),
               cand_idx=host.cand[:m].copy(), verdict=host.verdict[:m].copy(), truncated=False)
    dets = [(int(raw["classes"][i]), raw["boxes"][i], float(raw["scores"][i])) for i in host.pick[:r.n_det]]
    if keep is not None:
        cells = r.dims[0] * r.dims[1] * r.dims[2]
        view = lambda ptr, shape, dt=torch.float32: _device_view(ptr, shape, dt, dev, ws)
        X, Y, Z = r.dims
        keep.update(y=y, dims=(X, Y, Z), corner=tuple(r.corner), level_rows=list(r.level_rows),
                    grids=(view(r.d_grid_obj, (X, Y, Z)), view(r.d_grid_rot, (X, Y, Z, 2)), view(r.d_grid_scale, (X, Y, Z, 3))),
                    net_pred=(view(r.d_xyz, (n, 3)), view(r.d_scale, (n, 3)), view(r.d_prob, (n,)),
                              view(r.d_class, (n,), torch.int32)), raw=raw)
    return dets, raw, y


def last_scene_host_us(dev=None):
    """(plan + wait, network enqueue, head + vote enqueue, decode + wait) host microseconds of the calling thread's last
    detect_scene_c call on its current stream"""
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else dev
    host = _scene_hosts.get((dev.index, torch.cuda.current_stream(dev).cuda_stream))
    return host.last_host_us if host is not None else None


def _device_view(ptr, shape, dtype, dev, owner):
    """tensor view of a region of the scene scratch (valid until the next scene call on the same stream)"""
    import numpy as np
    count = int(np.prod(shape))
    off = int(ptr) - owner.data_ptr()
    nbytes = count * torch.empty((), dtype=dtype).element_size()
    assert 0 <= off and off + nbytes <= owner.numel()
    return owner[off:off + nbytes].view(dtype).view(*shape).clone()
