"""Drop-in for the reference's pybind module ``hv_cuda`` (houghvoting/src/hv_cuda.cpp:74-77).

    forward(points, xyz_labels, scale_labels, obj_labels, res, num_rots[, corners])
        -> [grid_obj[X,Y,Z], grid_rot[X,Y,Z,2], grid_scale[X,Y,Z,3]]      (hv_cuda.cpp:30-45)
    backward(grad_grid, points, xyz_labels, scale_labels, obj_labels, res, num_rots)
        -> [d_xyz_labels, d_scale_labels, d_obj_labels]                    (hv_cuda.cpp:47-71)

Same argument meaning, same input checks and messages (``"<name> must be a CUDA tensor"``,
``"<name> must be contiguous"``, hv_cuda.cpp:26-28), same fresh writable outputs on the inputs'
device.  The optional 7th ``corners[2,3]`` is the SUN RGB-D caller's variant
(sunrgbd/brnetcanon.py:99).  Under PyTorch-ROCm "cuda" tensors are HIP device memory; the
work is done by the hand-written gfx950 kernels in libcvhip.so on torch's current stream.

Differences that are improvements, not behaviour changes: kernels run on the current stream
instead of the legacy default stream, one host sync per forward (the grid shape has to reach
the host) instead of twelve, zero when ``corners`` is given.
"""
import ctypes
import threading

import torch

from . import _lib

ALGO_AUTO, ALGO_DIRECT, ALGO_TILES = 0, 1, 2
_algo = ALGO_AUTO


def set_algorithm(algo):
    """0 auto, 1 direct global atomics, 2 LDS tiles (for A/B measurements)."""
    global _algo
    _algo = int(algo)


def _remember_corner(grid_obj, mn):
    # kept on the tensor OBJECT (like _scalar's host value): a cache keyed by address would hand a grid that was
    # not produced by forward() (a loaded / golden grid on a recycled allocation of the same shape) a stale origin
    grid_obj._cv_corner = [float(v) for v in mn]


def recent_corner(grid_obj):
    """grid origin of a grid_obj returned by forward() (None for any other tensor)"""
    return getattr(grid_obj, "_cv_corner", None)


def _check_input(x, name):
    if not x.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not x.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _scalar(t, kind):
    """Host value of a 0-dim device tensor (res / num_rots).  The reference dereferences these
    on the device (hv_cuda_kernel.cu:22-23); we need them on the host for the launch geometry,
    so the value is read once per tensor OBJECT and version and kept on the object itself
    (a cache keyed by address would hand a recycled allocation the previous tensor's value)."""
    hit = getattr(t, "_cv_host_value", None)
    if hit is None or hit[0] != t._version:
        hit = (t._version, float(t.item()) if kind == "f" else int(t.item()))
        try:
            t._cv_host_value = hit
        except AttributeError:      # exotic tensor subclass without a __dict__: just re-read every call
            pass
    return hit[1]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f3(vals):
    return (ctypes.c_float * 3)(*[float(v) for v in vals])


_prefetched = {}
_pinned_pool = []
_prefetch_lock = threading.Lock()      # scene threads share the two containers (bench.py, pipeline.detect_scene)


def _pinned6():
    # pinned landing buffers are recycled: allocating page-locked memory costs about a millisecond per call
    with _prefetch_lock:
        if _pinned_pool:
            return _pinned_pool.pop()
    return torch.empty(6, dtype=torch.float32).pin_memory()


def _recycle(host):
    with _prefetch_lock:
        if len(_pinned_pool) < 64:
            _pinned_pool.append(host)


def reserve_pinned(count):
    """create `count` pinned landing buffers now (bench.py does it before the clock starts)"""
    bufs = [torch.empty(6, dtype=torch.float32).pin_memory() for _ in range(count)]
    for b in bufs:
        _recycle(b)


def prefetch_geometry(points):
    """Start the bounds reduction of `points` now, without waiting for it: a later forward()/grid_geometry() on
    the same (unmodified) tensor picks the result up instead of reducing and stalling in front of the vote.
    Call it before the network forward of the scene (pipeline.detect_scene does)."""
    L = _lib.lib()
    dev = points.device
    ws = _lib.scratch(dev, "hv_minmax", L.cv_hv_minmax_workspace_bytes())
    host = _pinned6()
    with torch.cuda.device(dev):
        _lib.check(L.cv_hv_minmax_async_f32(_ptr(points), points.shape[0], ctypes.c_void_p(host.data_ptr()),
                                            _ptr(ws), ws.numel(), _stream(dev)), "cv_hv_minmax_async_f32")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    stale = []
    with _prefetch_lock:
        # the entry keeps `points` alive, so the address cannot be recycled while it is cached
        # (keyed per host thread: two scene threads may work on the same resident scene at the same time)
        key = (threading.get_ident(), points.data_ptr())
        old = _prefetched.pop(key, None)
        _prefetched[key] = (points, points._version, host, ev)
        if old is not None:
            stale.append(old)
        if len(_prefetched) > 64:
            # entries nobody came back for (a prefetch without a vote): drop the oldest ones whose copy has landed -
            # torch's pinned allocator does not know about the raw hipMemcpyAsync, so a buffer released earlier could
            # be reissued and overwritten by a copy still in flight; entries of other threads still in flight stay
            for k in list(_prefetched)[:-32]:
                if _prefetched[k][3].query():
                    stale.append(_prefetched.pop(k))
    for ent in stale:
        ent[3].synchronize()
        _recycle(ent[2])


def _take_prefetched(points):
    with _prefetch_lock:
        hit = _prefetched.pop((threading.get_ident(), points.data_ptr()), None)
    if hit is None:
        return None
    hit[3].synchronize()
    if hit[0] is not points or hit[1] != points._version:
        _recycle(hit[2])
        return None
    h = hit[2].tolist()
    _recycle(hit[2])
    return h[:3], h[3:]


def grid_geometry(points, res):
    """(corner[3], max[3], dims[3]) as hv_cuda_kernel.cu:129-134 computes them (one host sync, or none when
    prefetch_geometry(points) ran earlier)."""
    L = _lib.lib()
    pre = _take_prefetched(points)
    if pre is not None:
        mn, mx = _f3(pre[0]), _f3(pre[1])
        dims = (ctypes.c_int * 3)()
        _lib.check(L.cv_hv_grid_dims_f32(mn, mx, ctypes.c_float(res), dims), "cv_hv_grid_dims_f32")
        return list(mn), list(mx), [int(d) for d in dims]
    ws = _lib.scratch(points.device, "hv_minmax", L.cv_hv_minmax_workspace_bytes())
    mn = (ctypes.c_float * 3)()
    mx = (ctypes.c_float * 3)()
    with torch.cuda.device(points.device):
        _lib.check(L.cv_hv_minmax_f32(_ptr(points), points.shape[0], mn, mx, _ptr(ws), ws.numel(),
                                      _stream(points.device)), "cv_hv_minmax_f32")
    dims = (ctypes.c_int * 3)()
    _lib.check(L.cv_hv_grid_dims_f32(mn, mx, ctypes.c_float(res), dims), "cv_hv_grid_dims_f32")
    return list(mn), list(mx), [int(d) for d in dims]


def _checked_inputs(points, xyz_labels, scale_labels, obj_labels, res, num_rots):
    _check_input(points, "points")
    _check_input(xyz_labels, "xyz_labels")
    _check_input(scale_labels, "scale_labels")
    _check_input(obj_labels, "obj_labels")
    _check_input(res, "res")
    _check_input(num_rots, "num_rots")
    for t, name in ((points, "points"), (xyz_labels, "xyz_labels"), (scale_labels, "scale_labels"),
                    (obj_labels, "obj_labels")):
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be float32 (got %s)" % (name, t.dtype))
    n = points.shape[0]
    if points.dim() != 2 or points.shape[1] != 3 or xyz_labels.shape != (n, 3) \
            or scale_labels.shape != (n, 3) or obj_labels.shape != (n,):
        raise RuntimeError("expected points/xyz_labels/scale_labels [N,3] and obj_labels [N]")
    if n == 0:
        # the reference fails inside torch::min on an empty tensor (hv_cuda_kernel.cu:129)
        raise RuntimeError("hv_cuda: cannot vote with zero points")
    return n, _scalar(res, "f"), _scalar(num_rots, "i")


def forward(points, xyz_labels, scale_labels, obj_labels, res, num_rots, corners=None):
    n, res_v, nrot = _checked_inputs(points, xyz_labels, scale_labels, obj_labels, res, num_rots)
    L = _lib.lib()
    dev = points.device
    if corners is None:
        mn, _, dims = grid_geometry(points, res_v)
    else:
        c = corners.detach().to("cpu", torch.float32)
        mn = [float(v) for v in c[0]]
        d = (ctypes.c_int * 3)()
        _lib.check(L.cv_hv_grid_dims_f32(_f3(mn), _f3(c[1]), ctypes.c_float(res_v), d),
                   "cv_hv_grid_dims_f32")
        dims = [int(v) for v in d]
    X, Y, Z = dims
    grid_obj = torch.empty((X, Y, Z), dtype=torch.float32, device=dev)
    grid_rot = torch.empty((X, Y, Z, 2), dtype=torch.float32, device=dev)
    grid_scale = torch.empty((X, Y, Z, 3), dtype=torch.float32, device=dev)
    cdims = (ctypes.c_int * 3)(*dims)
    wsb = L.cv_hv_forward_workspace_bytes(n, nrot, cdims, _algo)
    ws = _lib.scratch(dev, "hv_forward", wsb)
    with torch.cuda.device(dev):
        _lib.check(L.cv_hv_forward_f32(_ptr(points), _ptr(xyz_labels), _ptr(scale_labels),
                                       _ptr(obj_labels), n, ctypes.c_float(res_v), nrot, _f3(mn),
                                       cdims, _ptr(grid_obj), _ptr(grid_rot), _ptr(grid_scale),
                                       _ptr(ws), ws.numel(), _algo, _stream(dev)),
                   "cv_hv_forward_f32")
    # remember the grid origin of this grid so the decode that follows does not reduce the points again
    _remember_corner(grid_obj, mn)
    return [grid_obj, grid_rot, grid_scale]


def backward(grad_grid, points, xyz_labels, scale_labels, obj_labels, res, num_rots, corners=None):
    """``corners`` (optional, not in the reference's 7-positional signature): the [2,3] box the forward was
    given; the grid origin is then corners[0] as in forward() instead of the minimum of the points."""
    _check_input(grad_grid, "grad_grid")
    n, res_v, nrot = _checked_inputs(points, xyz_labels, scale_labels, obj_labels, res, num_rots)
    if grad_grid.dtype != torch.float32 or grad_grid.dim() != 3:
        raise RuntimeError("grad_grid must be a float32 [X,Y,Z] tensor")
    L = _lib.lib()
    dev = points.device
    if corners is None:
        mn, _, _ = grid_geometry(points, res_v)      # hv_cuda_kernel.cu:274-276
    else:
        mn = [float(v) for v in corners.detach().to("cpu", torch.float32)[0]]
    cdims = (ctypes.c_int * 3)(*grad_grid.shape)     # :200 sizes come from grad_grid
    d_xyz = torch.empty_like(xyz_labels)
    d_scale = torch.empty_like(scale_labels)
    d_obj = torch.empty_like(obj_labels)
    with torch.cuda.device(dev):
        _lib.check(L.cv_hv_backward_f32(_ptr(grad_grid), _ptr(points), _ptr(xyz_labels),
                                        _ptr(scale_labels), _ptr(obj_labels), n,
                                        ctypes.c_float(res_v), nrot, _f3(mn), cdims, _ptr(d_xyz),
                                        _ptr(d_scale), _ptr(d_obj), _stream(dev)),
                   "cv_hv_backward_f32")
    return [d_xyz, d_scale, d_obj]


def count_votes(points, xyz_labels, scale_labels, res, num_rots, corner, dims):
    """Number of in-bounds (point, rotation) votes: V_in of the algorithmic byte count."""
    L = _lib.lib()
    dev = points.device
    ws = _lib.scratch(dev, "hv_count", 256)
    out = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(L.cv_hv_count_votes_f32(_ptr(points), _ptr(xyz_labels), _ptr(scale_labels),
                                           points.shape[0], ctypes.c_float(res), int(num_rots),
                                           _f3(corner), (ctypes.c_int * 3)(*dims), ctypes.byref(out),
                                           _ptr(ws), ws.numel(), _stream(dev)),
                   "cv_hv_count_votes_f32")
    return int(out.value)
