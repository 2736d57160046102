"""Seeded synthetic ScanNet-shaped scenes.

Reproduces only the OUTPUT CONTRACT of the reference data loader
(utils/dataloader.py:170-210): per active 3 cm voxel -> integer coords
``floor(p/res)``, rgb in [0,1], LCC label ``xyz`` (point in the box frame divided
by the half extents), ``scale`` label (half extents), ``class`` in 0..8 or 9 for
background.  The y axis is up and every rotation is about y with
R_y(t) = [[cos,0,-sin],[0,1,0],[sin,0,cos]] (dataloader.py:163-164,
eval_joint.py:215).

There is no network and no ScanNet here, so bench.py / smoke() / the tests all
run on these scenes (``"data": "synthetic"``).
"""
from dataclasses import dataclass

import numpy as np

NUM_CLASSES = 9
BACKGROUND = 9


@dataclass
class Scene:
    seed: int
    res: float
    coords: np.ndarray        # [N,3] int32 voxel coordinates (already unique)
    feats: np.ndarray         # [N,3] float32 rgb in [0,1]
    xyz_labels: np.ndarray    # [N,3] float32
    scale_labels: np.ndarray  # [N,3] float32 half extents (0 for background)
    class_labels: np.ndarray  # [N]   int32, 9 = background
    boxes: np.ndarray         # [K,8] float32: cx,cy,cz,yaw,sx,sy,sz,class (ground truth, metres)

    @property
    def points(self):
        """World-space points exactly as eval_joint.py:193 forms them: int coords * res in fp32."""
        return (self.coords.astype(np.float32) * np.float32(self.res)).astype(np.float32)


def _ry(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]], np.float64)


def _sample_rect(rng, n, origin, u, v):
    a = rng.random((n, 1))
    b = rng.random((n, 1))
    return origin[None] + a * u[None] + b * v[None]


def make_scene(seed, n_points=80000, res=0.03, room=(5.2, 2.6, 5.2), n_boxes=12,
               origin_shift=True, margin=1.0, box_scale=1.0):
    """SURVEY 8d generator: room shell + K oriented boxes on the floor -> exactly n_points voxels."""
    rng = np.random.default_rng(seed)
    W, H, D = room
    # --- objects -------------------------------------------------------------
    # Boxes are placed one at a time without overlap (bounding circles in the xz plane):
    # overlapping boxes put foreign points inside each other's volume and the reference's
    # back-projection check (eval_joint.py:249-253) then rejects every candidate, which
    # would make the decode stage of the benchmark meaningless.  A box that cannot be
    # placed after 60 tries is shrunk by 0.85 and tried again (keeps K fixed).
    half = np.zeros((n_boxes, 3))
    yaw = rng.uniform(0, 2 * np.pi, n_boxes)
    ctr = np.zeros((n_boxes, 3))
    for b in range(n_boxes):
        h = np.array([rng.uniform(0.25, 0.9), rng.uniform(0.3, 0.6), rng.uniform(0.25, 0.9)]) * box_scale
        for attempt in range(2000):
            if attempt and attempt % 60 == 0:
                h[[0, 2]] *= 0.85
            rad = np.hypot(h[0], h[2])
            m = min(margin, max(rad + 0.05, 0.0))
            c = np.array([rng.uniform(m, max(W - m, m + 1e-6)), h[1], rng.uniform(m, max(D - m, m + 1e-6))])
            if b == 0:
                break
            prev_rad = np.hypot(half[:b, 0], half[:b, 2])
            gap = np.hypot(ctr[:b, 0] - c[0], ctr[:b, 2] - c[2]) - prev_rad - rad
            if np.all(gap > 0.06):
                break
        half[b], ctr[b] = h, c
    cls = rng.integers(0, NUM_CLASSES, n_boxes)
    # --- surfaces: (area, sampler) -------------------------------------------
    surfs = []   # (area, origin, u, v, box_id or -1)
    z3 = np.zeros(3)
    surfs.append((W * D, z3, np.array([W, 0, 0.]), np.array([0, 0, D]), -1))            # floor
    surfs.append((W * H, z3, np.array([W, 0, 0.]), np.array([0, H, 0.]), -1))           # wall z=0
    surfs.append((W * H, np.array([0, 0, D]), np.array([W, 0, 0.]), np.array([0, H, 0.]), -1))
    surfs.append((D * H, z3, np.array([0, 0, D]), np.array([0, H, 0.]), -1))            # wall x=0
    surfs.append((D * H, np.array([W, 0, 0.]), np.array([0, 0, D]), np.array([0, H, 0.]), -1))
    for b in range(n_boxes):
        hx, hy, hz = half[b]
        # 5 visible faces in the box frame (no bottom), as (origin,u,v) in units of half extents
        faces = [((-1, 1, -1), (2, 0, 0), (0, 0, 2)),    # top
                 ((-1, -1, -1), (2, 0, 0), (0, 2, 0)), ((-1, -1, 1), (2, 0, 0), (0, 2, 0)),
                 ((-1, -1, -1), (0, 0, 2), (0, 2, 0)), ((1, -1, -1), (0, 0, 2), (0, 2, 0))]
        for o, u, v in faces:
            o = np.array(o, float) * half[b]
            u = np.array(u, float) * half[b]
            v = np.array(v, float) * half[b]
            surfs.append((np.linalg.norm(np.cross(u, v)), o, u, v, b))
    areas = np.array([s[0] for s in surfs])
    prob = areas / areas.sum()
    colors = rng.uniform(0, 1, (len(surfs), 3))

    def draw(m):
        which = rng.choice(len(surfs), size=m, p=prob)
        pts = np.zeros((m, 3))
        xyz = np.zeros((m, 3))
        sc = np.zeros((m, 3))
        cl = np.full(m, BACKGROUND, np.int32)
        bid = np.full(m, -1, np.int64)
        rgb = colors[which] + rng.normal(0, 0.05, (m, 3))
        for si, (_, o, u, v, b) in enumerate(surfs):
            sel = np.nonzero(which == si)[0]
            if sel.size == 0:
                continue
            local = _sample_rect(rng, sel.size, o, u, v)
            if b < 0:
                pts[sel] = local
            else:
                # world point of an object point: p = c + R_y(yaw) (s o x)   
                pts[sel] = ctr[b][None] + local @ _ry(yaw[b]).T
                sc[sel] = half[b][None]
                cl[sel] = cls[b]
                bid[sel] = b
        pts += rng.normal(0, 0.005, pts.shape)
        # labels follow the jittered point like the reference (inverse box transform of the point)
        for b in range(n_boxes):
            sel = np.nonzero(bid == b)[0]
            if sel.size:
                xyz[sel] = ((pts[sel] - ctr[b][None]) @ _ry(yaw[b])) / half[b][None]
        return pts, xyz, sc, cl, np.clip(rgb, 0, 1)

    need = n_points
    P, Xl, Sl, Rg = np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3))
    Cl = np.zeros(0, np.int32)
    vox = np.zeros((0, 3), np.int64)
    m = int(need * 3)
    for _ in range(12):
        p, x, s, c, g = draw(m)
        P = np.concatenate([P, p]); Xl = np.concatenate([Xl, x]); Sl = np.concatenate([Sl, s])
        Cl = np.concatenate([Cl, c]); Rg = np.concatenate([Rg, g])
        vox = np.floor(P / res).astype(np.int64)
        # first sample per voxel, like ME.utils.sparse_quantize(return_index=True)
        _, first = np.unique(vox, axis=0, return_index=True)
        if first.size >= need:
            break
        m = int(need * 2)
    first = np.sort(first)
    if first.size >= need:
        first = np.sort(rng.choice(first, size=need, replace=False))
    else:
        raise ValueError("scene too small for n_points=%d (got %d voxels)" % (need, first.size))
    coords = vox[first]
    if origin_shift:
        coords = coords + rng.integers(-200, 201, 3)[None]
    gt = np.concatenate([ctr, yaw[:, None], half, cls[:, None].astype(float)], -1)
    if origin_shift:
        gt = gt.copy()
        gt[:, :3] += (coords[0] - vox[first][0])[None] * res
    return Scene(seed=seed, res=res, coords=coords.astype(np.int32), feats=Rg[first].astype(np.float32),
                 xyz_labels=Xl[first].astype(np.float32), scale_labels=Sl[first].astype(np.float32),
                 class_labels=Cl[first].astype(np.int32), boxes=gt.astype(np.float32))


def synth_predictions(scene, seed=None):
    """Per-point network outputs synthesised from the labels (vote/decode-only runs).

    object points:  xyz = label + N(0,0.05), scale = label*exp(N(0,0.05)), prob ~ U(0.7,1)
    background:     xyz ~ U(-1,1)^3,         scale ~ U(0.2,0.8),          prob ~ U(0,0.1)
    Returns xyz[N,3], scale[N,3], prob[N] (float32) and class[N] (int32, in 0..8)."""
    rng = np.random.default_rng(1000003 + (scene.seed if seed is None else seed))
    n = scene.coords.shape[0]
    is_obj = scene.class_labels != BACKGROUND
    xyz = rng.uniform(-1, 1, (n, 3))
    scale = rng.uniform(0.2, 0.8, (n, 3))
    prob = rng.uniform(0, 0.1, n)
    cls = rng.integers(0, NUM_CLASSES, n)
    k = int(is_obj.sum())
    xyz[is_obj] = scene.xyz_labels[is_obj] + rng.normal(0, 0.05, (k, 3))
    scale[is_obj] = scene.scale_labels[is_obj] * np.exp(rng.normal(0, 0.05, (k, 3)))
    prob[is_obj] = rng.uniform(0.7, 1.0, k)
    cls[is_obj] = scene.class_labels[is_obj]
    return (xyz.astype(np.float32), scale.astype(np.float32), prob.astype(np.float32),
            cls.astype(np.int32))
