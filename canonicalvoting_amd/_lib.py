"""ctypes loader for the C-ABI library (include/cv_hip.h).

There is no CPU fallback: if the HIP library has not been built the import of any
op fails loudly with instructions, and every entry point checks its return code.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_C", "libcvhip.so")

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
vp = ctypes.c_void_p
ABI_VERSION = 3          # CV_ABI_VERSION of include/cv_hip.h that these ctypes signatures were written against


class CvError(RuntimeError):
    pass


class DecodeParams(ctypes.Structure):
    """struct cv_decode_params (include/cv_hip.h); defaults = eval_joint.py:18-21,245,252."""
    _fields_ = [("thresh_high", ctypes.c_float), ("thresh_low", ctypes.c_float),
                ("valid_ratio", ctypes.c_float), ("elimination", ctypes.c_int),
                ("prob_thresh", ctypes.c_float), ("elim_hi_plus1", ctypes.c_int),
                ("max_iters", ctypes.c_int), ("err_thresh", ctypes.c_double)]


class ConvDesc(ctypes.Structure):
    """struct cv_conv_desc (include/cv_hip.h)"""
    _fields_ = [("in_", vp), ("n_in", ctypes.c_longlong), ("in_ld", ctypes.c_int), ("cin", ctypes.c_int),
                ("weight", vp), ("K", ctypes.c_int), ("cout", ctypes.c_int), ("nbr", vp),
                ("n_out", ctypes.c_longlong), ("scale", vp), ("shift", vp), ("residual", vp),
                ("res_ld", ctypes.c_int), ("relu", ctypes.c_int), ("out", vp), ("out_ld", ctypes.c_int),
                ("flavour", ctypes.c_int), ("ws", vp), ("ws_bytes", ctypes.c_size_t), ("row_perm", vp),
                ("j_begin", ctypes.c_int), ("j_end", ctypes.c_int), ("acc_in", vp), ("acc_ld", ctypes.c_int),
                ("perm_groups", ctypes.c_int), ("plan_ent", vp), ("plan_cnt", vp),
                ("weight_packed", vp), ("weight_x6", vp), ("in2", vp), ("in2_ld", ctypes.c_int), ("cin2", ctypes.c_int),
                ("weight2_x6", vp), ("perm_has_map", ctypes.c_int), ("weight_pieces", ctypes.c_int),
                ("acc_scale", ctypes.c_float), ("range_flag", vp), ("in_hl", ctypes.c_int), ("out_hl", ctypes.c_int),
                ("res_hl", ctypes.c_int), ("split_tickets", vp), ("acc_scale_dev", vp)]


class SceneMaps(ctypes.Structure):
    """struct cv_scene_maps (include/cv_hip.h)"""
    _fields_ = [("stem", ctypes.c_longlong), ("out", ctypes.c_longlong), ("down", ctypes.c_longlong * 4),
                ("k3", ctypes.c_longlong * 5), ("up", ctypes.c_longlong * 4), ("mask_perm", ctypes.c_longlong * 5),
                ("up_perm", ctypes.c_longlong * 4), ("scratch", ctypes.c_longlong), ("bitmap", ctypes.c_longlong)]


class NetBuf(ctypes.Structure):
    """struct cv_net_buf (include/cv_hip.h)"""
    _fields_ = [("level", ctypes.c_int), ("channels", ctypes.c_int), ("rows_level", ctypes.c_int), ("hl", ctypes.c_int)]


class NetOp(ctypes.Structure):
    """struct cv_net_op (include/cv_hip.h)"""
    _fields_ = [("in_buf", ctypes.c_int), ("in_col", ctypes.c_int), ("cin", ctypes.c_int),
                ("out_buf", ctypes.c_int), ("out_col", ctypes.c_int), ("cout", ctypes.c_int),
                ("res_buf", ctypes.c_int), ("res_col", ctypes.c_int), ("map", ctypes.c_int), ("K", ctypes.c_int),
                ("perm", ctypes.c_int), ("perm_groups", ctypes.c_int), ("relu", ctypes.c_int),
                ("weight", vp), ("scale", vp), ("shift", vp), ("weight_x6", vp), ("in2_buf", ctypes.c_int),
                ("in2_col", ctypes.c_int), ("cin2", ctypes.c_int), ("weight2_x6", vp), ("weight_pieces", ctypes.c_int), ("acc_scale", ctypes.c_float)]


class SceneDesc(ctypes.Structure):
    """struct cv_scene_desc (include/cv_hip.h)"""
    _fields_ = [("d_coords4", vp), ("n", ctypes.c_longlong), ("d_feats", vp), ("feats_ld", ctypes.c_int), ("d_points", vp),
                ("res", ctypes.c_float), ("num_rots", ctypes.c_int), ("ops", vp), ("n_ops", ctypes.c_int), ("bufs", vp),
                ("n_bufs", ctypes.c_int), ("stem_k", ctypes.c_int), ("mask_groups", ctypes.c_int),
                ("masked_min_rows", ctypes.c_longlong), ("max_channels", ctypes.c_int),
                ("use_range_flag", ctypes.c_int),
                ("d_out_feats", vp), ("out_ld", ctypes.c_int), ("out_channels", ctypes.c_int), ("nclasses", ctypes.c_int),
                ("log_scale", ctypes.c_int), ("d_xyz_in", vp), ("d_scale_in", vp), ("d_prob_in", vp), ("d_class_in", vp),
                ("vote_algo", ctypes.c_int), ("decode", DecodeParams), ("max_candidates", ctypes.c_int),
                ("nms_threshold", ctypes.c_double), ("d_ws", vp), ("ws_bytes", ctypes.c_size_t), ("d_grids", vp),
                ("grid_capacity_floats", ctypes.c_size_t), ("h_pinned", vp), ("pinned_bytes", ctypes.c_size_t),
                ("h_cand_idx", vp), ("h_verdict", vp), ("h_boxes", vp), ("h_scores", vp), ("h_classes", vp), ("h_pick", vp),
                ("adaptive_split", ctypes.c_int), ("events", vp * 5),
                ("conv_split_target", ctypes.c_int), ("vote_part_records", ctypes.c_int)]


class PackJob(ctypes.Structure):
    """struct cv_pack_job (include/cv_hip.h)"""
    _fields_ = [("w", vp), ("wp", vp), ("K", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("trans", ctypes.c_int),
                ("scale_log2", ctypes.c_int), ("reserved", ctypes.c_int)]


class SceneResult(ctypes.Structure):
    """struct cv_scene_result (include/cv_hip.h)"""
    _fields_ = [("n_cand", ctypes.c_int), ("n_boxes", ctypes.c_int), ("n_det", ctypes.c_int), ("truncated", ctypes.c_int),
                ("range_flag", ctypes.c_int), ("duplicates", ctypes.c_int), ("out_of_window", ctypes.c_int),
                ("scenes_in_flight", ctypes.c_int), ("dims", ctypes.c_int * 3), ("corner", ctypes.c_float * 3), ("level_rows", ctypes.c_longlong * 5),
                ("needed_ws_bytes", ctypes.c_size_t), ("needed_grid_floats", ctypes.c_size_t),
                ("d_grid_obj", vp), ("d_grid_rot", vp), ("d_grid_scale", vp), ("d_xyz", vp), ("d_scale", vp),
                ("d_prob", vp), ("d_class", vp), ("host_us", ctypes.c_float * 4)]


# symbol -> (restype, argtypes); tests check every symbol of include/cv_hip.h is here and exported
SIGNATURES = {
    "cv_abi_version": (ctypes.c_int, []),
    "cv_last_error": (ctypes.c_char_p, []),
    "cv_hv_minmax_workspace_bytes": (ctypes.c_size_t, []),
    "cv_hv_minmax_f32": (ctypes.c_int, [vp, ctypes.c_int64, c_float_p, c_float_p, vp, ctypes.c_size_t, vp]),
    "cv_hv_minmax_async_f32": (ctypes.c_int, [vp, ctypes.c_int64, vp, vp, ctypes.c_size_t, vp]),
    "cv_hv_grid_dims_f32": (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_float, c_int_p]),
    "cv_hv_forward_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, c_int_p, ctypes.c_int]),
    "cv_hv_forward_f32": (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                         c_float_p, c_int_p, vp, vp, vp, vp, ctypes.c_size_t,
                                         ctypes.c_int, vp]),
    "cv_hv_set_kernel_events": (ctypes.c_int, [vp, vp]),
    "cv_hv_backward_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_float,
                                          ctypes.c_int, c_float_p, c_int_p, vp, vp, vp, vp]),
    "cv_hv_count_votes_f32": (ctypes.c_int, [vp, vp, vp, ctypes.c_int64, ctypes.c_float, ctypes.c_int,
                                             c_float_p, c_int_p, c_i64_p, vp, ctypes.c_size_t, vp]),
    "cv_decode_workspace_bytes": (ctypes.c_size_t, [c_int_p, ctypes.c_int64, ctypes.c_int]),
    "cv_decode_f32": (ctypes.c_int, [vp, vp, vp, c_int_p, c_float_p, ctypes.c_float, vp, vp, vp, vp,
                                     ctypes.c_int64, ctypes.POINTER(DecodeParams), ctypes.c_int, vp,
                                     ctypes.c_size_t, c_int_p, c_i64_p, c_i32_p, c_int_p, c_float_p,
                                     c_float_p, c_i32_p, c_int_p, vp]),
    "cv_sp_table_capacity": (ctypes.c_longlong, [ctypes.c_longlong]),
    "cv_sp_levels_workspace_bytes": (ctypes.c_size_t, [ctypes.c_longlong]),
    "cv_sp_build_levels": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                          ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, vp,
                                          c_i32_p, vp, ctypes.c_size_t, vp]),
    "cv_sp_morton_keys": (ctypes.c_int, [vp, ctypes.c_longlong, vp, vp]),
    "cv_sp_sort_workspace_bytes": (ctypes.c_size_t, [ctypes.c_longlong]),
    "cv_sp_sort_rows": (ctypes.c_int, [vp, ctypes.c_longlong, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "cv_sp_pack_weights_stem_h2_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "cv_sp_to_hl_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "cv_sp_from_hl_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]),
    "cv_sp_kernel_map": (ctypes.c_int, [vp, ctypes.c_longlong, vp, vp, ctypes.c_longlong, ctypes.c_int,
                                        ctypes.c_int, vp, vp]),
    "cv_sp_up_map": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_longlong, vp, vp]),
    "cv_sp_conv_workspace_bytes": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]),
    "cv_sp_conv_f32": (ctypes.c_int, [ctypes.POINTER(ConvDesc), vp]),
    "cv_sp_set_split_target": (ctypes.c_int, [ctypes.c_int]),
    "cv_hv_set_part_records": (ctypes.c_int, [ctypes.c_int]),
    "cv_hv_set_part_records_thread": (ctypes.c_int, [ctypes.c_int]),
    "cv_sp_pack_weights_h2_batch_f32": (ctypes.c_int, [ctypes.POINTER(PackJob), ctypes.c_int, vp, vp]),
    "cv_sp_copy_unless_flag": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_longlong), ctypes.c_int, vp, vp]),
    "cv_sp_pack_weights_t_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "cv_sp_set_split_target_thread": (ctypes.c_int, [ctypes.c_int]),
    "cv_sp_set_ablation": (ctypes.c_int, [ctypes.c_int]),
    "cv_sp_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "cv_sp_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]),
    "cv_sp_pack_weights_x6_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "cv_sp_pack_weights_h2_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]),
    "cv_sp_pack_weights_bf16_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "cv_sp_scene_maps_words": (ctypes.c_size_t, [c_i64_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_longlong, ctypes.POINTER(SceneMaps)]),
    "cv_sp_scene_maps": (ctypes.c_int, [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_longlong,
                                        c_i64_p, vp, ctypes.c_longlong, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_longlong, vp, ctypes.c_size_t, vp]),
    "cv_sp_scene_plan_words": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_longlong]),
    "cv_sp_scene_plan": (ctypes.c_int, [vp, ctypes.c_longlong, vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                        ctypes.POINTER(vp), ctypes.c_longlong, vp, c_i32_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_longlong, vp, ctypes.c_size_t, ctypes.POINTER(SceneMaps), vp,
                                        ctypes.c_size_t, vp, ctypes.c_size_t, vp]),
    "cv_net_arena_bytes": (ctypes.c_size_t, [ctypes.POINTER(NetBuf), ctypes.c_int, c_i64_p, ctypes.c_int]),
    "cv_net_run_f32": (ctypes.c_int, [ctypes.POINTER(NetOp), ctypes.c_int, ctypes.POINTER(NetBuf), ctypes.c_int,
                                      c_i64_p, ctypes.c_int, vp, ctypes.c_size_t, ctypes.POINTER(vp), c_int_p,
                                      ctypes.POINTER(vp), ctypes.c_int, ctypes.POINTER(vp), ctypes.c_int,
                                      vp, ctypes.c_size_t, vp, vp]),
    "cv_sp_mask_keys": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "cv_sp_mask_perms": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_size_t,
                                        ctypes.c_int, vp]),
    "cv_sp_transpose_map": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, vp, vp]),
    "cv_sp_wgrad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "cv_sp_conv_wgrad_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp,
                                            ctypes.c_int, ctypes.c_longlong, vp, vp, ctypes.c_size_t, vp]),
    "cv_sp_conv_wgrad_px_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp,
                                               ctypes.c_int, ctypes.c_longlong, vp, vp, ctypes.c_size_t, ctypes.c_int,
                                               vp]),
    "cv_sp_col_sum_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp]),
    "cv_sp_col_sum_workspace_bytes": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int]),
    "cv_sp_col_sum_det_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_size_t, vp]),
    "cv_sp_bn_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "cv_sp_bn_stats_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_float,
                                          ctypes.c_float, vp, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "cv_sp_bn_backward_f32": (ctypes.c_int, [vp, vp, vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp,
                                             ctypes.c_float, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]),
    "cv_sp_bn_backward_hl_f32": (ctypes.c_int, [vp, vp, vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp,
                                                ctypes.c_float, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, vp, vp]),
    "cv_sp_affine_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                        ctypes.c_int, vp, ctypes.c_int, vp]),
    "cv_sp_affine_hl_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                           ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]),
    "cv_sp_bn_fold_f32": (ctypes.c_int, [vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_int, vp, vp, vp]),
    "cv_head_joint_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         vp, vp, vp, vp, vp]),
    "cv_head_separate_f32": (ctypes.c_int, [vp, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    "cv_detect_scene_f32": (ctypes.c_int, [ctypes.POINTER(SceneDesc), ctypes.POINTER(SceneResult), vp]),
    "cv_iou_obb": (ctypes.c_double, [c_float_p, c_float_p]),
    "cv_nms_obb": (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_double, c_i32_p]),
}

_lib = None


def lib():
    """The loaded libcvhip.so (raises CvError with build instructions when it is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CvError(
                "canonicalvoting_amd: HIP library %s not found. Build it with "
                "`python -m canonicalvoting_amd.csrc.build` (hipcc, gfx950). There is no CPU "
                "fallback for this op." % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime; it has to be the one already resident when
        # libcvhip.so resolves libamdhip64, otherwise two runtimes end up in the process and ours
        # sees no device ("no ROCm-capable device is detected").
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


_scratch_lock = threading.Lock()
_scratch_bufs = {}


def scratch(dev, name, nbytes):
    """Persistent device scratch of the CURRENT stream of `dev`, grown on demand: ``nbytes`` of uint8 named ``name``.

    Internal workspaces (the executor arena, sort / level / vote / decode workspaces) used to be ``torch.empty`` per
    call: every scene-thread stream owns a separate caching-allocator pool, so the first scenes of a stream - and any
    later scene whose sizes differ - paid a hipMalloc inside the timed region.  A workspace is consumed by launches
    of one stream only and the next call on that stream is ordered behind them, so one buffer per (device, stream,
    name) is safe; results handed to the caller are never scratch."""
    import torch
    key = (dev.index if dev.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(dev).cuda_stream, name)
    nbytes = max(int(nbytes), 256)
    with _scratch_lock:
        buf = _scratch_bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        # (the old buffer may still be read by queued launches: the caching allocator keeps its block stream-ordered)
        buf = torch.empty(nbytes + nbytes // 8, dtype=torch.uint8, device=dev)
        with _scratch_lock:
            _scratch_bufs[key] = buf
    return buf


def release_scratch(dev=None, stream=None):
    """Drop the persistent scratch buffers of one stream (``stream`` = a torch.cuda.Stream or a raw handle), of one
    device, or all of them (no arguments): the blocks go back to torch's caching allocator, where ``empty_cache()`` can
    free them.  Call it when a scene thread's stream is retired - scratch is keyed by the raw stream handle and would
    otherwise stay pinned for the life of the process (~40 MB per stream for 80k-point scenes)."""
    h = getattr(stream, "cuda_stream", stream)
    d = getattr(dev, "index", dev)
    with _scratch_lock:
        for key in [k for k in _scratch_bufs if (d is None or k[0] == d) and (h is None or k[1] == h)]:
            del _scratch_bufs[key]


def scratch_bytes():
    """bytes currently held by scratch(), for tests and the bench's memory report"""
    with _scratch_lock:
        return sum(b.numel() for b in _scratch_bufs.values())


def check(rc, what):
    if rc != 0:
        msg = lib().cv_last_error()
        raise CvError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
