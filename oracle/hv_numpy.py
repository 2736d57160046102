"""Independent numpy restatement of the reference vote op -- TEST INFRASTRUCTURE.

Cross-check for oracle/hv_oracle.c ("C restatement vs an independent
restatement").  Vectorised over (point, rotation); per-vote contributions are
formed in fp32 with the same operation order as hv_cuda_kernel.cu:29-59 and
accumulated in float64 with np.add.at, so it also serves as a high-precision
sum against which both the fp32-sequential C oracle and the fp32-atomic HIP
kernel can be measured.

Follows houghvoting/src/hv_cuda_kernel.cu:25-96 (vote), :106-118 (average),
:129-134 (grid dims).
"""
import numpy as np

f32 = np.float32


def grid_dims(points, res):
    pts = np.asarray(points, f32)
    mn = pts.min(0)
    mx = pts.max(0)
    diff = ((mx - mn).astype(f32) / f32(res)).astype(f32)     # :131 fp32 tensor ops
    dims = [int(np.trunc(d)) + 1 for d in diff]               # :132 .to<int>() + 1
    return mn, mx, dims


def rot_table(num_rots):
    rot_interval = f32(f32(2) * f32(3.141592654)) / f32(num_rots)   # :35
    theta = (np.arange(num_rots).astype(f32) * rot_interval).astype(f32)  # :37
    return np.cos(theta.astype(np.float64)).astype(f32), np.sin(theta.astype(np.float64)).astype(f32)


def hv_forward(points, xyz, scale, obj, res, num_rots, acc_dtype=np.float64):
    pts = np.asarray(points, f32)
    xyz = np.asarray(xyz, f32)
    scale = np.asarray(scale, f32)
    obj = np.asarray(obj, f32)
    res = f32(res)
    corner, _, dims = grid_dims(pts, res)
    X, Y, Z = dims
    ct, st = rot_table(num_rots)
    corr = (xyz * scale).astype(f32)                                   # :29-33
    cx, cy, cz = corr[:, 0:1], corr[:, 1:2], corr[:, 2:3]
    ox = ((-ct)[None] * cx).astype(f32) + (st[None] * cz).astype(f32)  # :38
    oy = np.broadcast_to(-cy, ox.shape)
    oz = ((-st)[None] * cx).astype(f32) - (ct[None] * cz).astype(f32)  # :39
    g = [(((pts[:, k:k + 1] + o).astype(f32) - corner[k]).astype(f32) / res).astype(f32)
         for k, o in enumerate((ox, oy, oz))]                          # :40
    ok = (g[0] >= 0) & (g[1] >= 0) & (g[2] >= 0) & (g[0] < f32(X - 1)) & (g[1] < f32(Y - 1)) \
        & (g[2] < f32(Z - 1))                                          # :41-44
    pi, ri = np.nonzero(ok)
    gx, gy, gz = (a[pi, ri] for a in g)
    fl = [np.trunc(a).astype(np.int64) for a in (gx, gy, gz)]          # :45
    fr = [(a - np.floor(a)).astype(f32) for a in (gx, gy, gz)]         # :47
    w0 = [(f32(1) - a).astype(f32) for a in fr]                        # :49
    w1 = fr                                                            # :50
    o = obj[pi]
    g_obj = np.zeros(X * Y * Z, acc_dtype)
    g_rot = np.zeros((X * Y * Z, 2), acc_dtype)
    g_scale = np.zeros((X * Y * Z, 3), acc_dtype)
    for bx in (0, 1):
        for by in (0, 1):
            for bz in (0, 1):
                wx = (w1 if bx else w0)[0]
                wy = (w1 if by else w0)[1]
                wz = (w1 if bz else w0)[2]
                w = (((wx * wy).astype(f32) * wz).astype(f32) * o).astype(f32)   # :52-59
                cell = ((fl[0] + bx) * Y + (fl[1] + by)) * Z + (fl[2] + bz)
                np.add.at(g_obj, cell, w.astype(acc_dtype))
                np.add.at(g_rot[:, 0], cell, (w * ct[ri]).astype(f32).astype(acc_dtype))
                np.add.at(g_rot[:, 1], cell, (w * st[ri]).astype(f32).astype(acc_dtype))
                for j in range(3):
                    np.add.at(g_scale[:, j], cell, (w * scale[pi, j]).astype(f32).astype(acc_dtype))
    # :112-117  x /= w + 1e-7 (double), stored as float.  The grid weight the
    # reference divides by is the fp32-accumulated one; mirror that rounding.
    w32 = g_obj.astype(f32)
    d = w32.astype(np.float64) + 1e-7
    g_rot = (g_rot.astype(f32).astype(np.float64) / d[:, None]).astype(f32)
    g_scale = (g_scale.astype(f32).astype(np.float64) / d[:, None]).astype(f32)
    return (w32.reshape(X, Y, Z), g_rot.reshape(X, Y, Z, 2), g_scale.reshape(X, Y, Z, 3),
            int(ok.sum()))


def contribution_counts(points, xyz, scale, res, num_rots, corner=None, dims=None, chunk=20000):
    """int32 [X,Y,Z]: how many (vote, corner) contributions each cell receives (hv_cuda_kernel.cu:41-96: eight per
    in-bounds vote).  Tests use it to state the accumulated-rounding bound of a cell's sums; same fp32 geometry as
    hv_forward above, chunked over points so that a 300k-point scene stays within a few hundred MB."""
    pts = np.asarray(points, f32)
    xyz = np.asarray(xyz, f32)
    scale = np.asarray(scale, f32)
    res = f32(res)
    if corner is None:
        corner, _, dims = grid_dims(pts, res)
    corner = np.asarray(corner, f32)
    X, Y, Z = dims
    ct, st = rot_table(num_rots)
    counts = np.zeros(X * Y * Z, np.int64)
    for a in range(0, len(pts), chunk):
        p = pts[a:a + chunk]
        corr = (xyz[a:a + chunk] * scale[a:a + chunk]).astype(f32)
        cx, cy, cz = corr[:, 0:1], corr[:, 1:2], corr[:, 2:3]
        ox = ((-ct)[None] * cx).astype(f32) + (st[None] * cz).astype(f32)
        oy = np.broadcast_to(-cy, ox.shape)
        oz = ((-st)[None] * cx).astype(f32) - (ct[None] * cz).astype(f32)
        g = [(((p[:, k:k + 1] + o).astype(f32) - corner[k]).astype(f32) / res).astype(f32)
             for k, o in enumerate((ox, oy, oz))]
        ok = (g[0] >= 0) & (g[1] >= 0) & (g[2] >= 0) & (g[0] < f32(X - 1)) & (g[1] < f32(Y - 1)) & (g[2] < f32(Z - 1))
        fl = [np.trunc(v[ok]).astype(np.int64) for v in g]
        base = (fl[0] * Y + fl[1]) * Z + fl[2]
        for bx in (0, 1):
            for by in (0, 1):
                for bz in (0, 1):
                    counts += np.bincount(base + (bx * Y + by) * Z + bz, minlength=X * Y * Z)
    return counts.reshape(X, Y, Z).astype(np.int32)
