"""Golden vectors for the data contract, produced by running the reference's own
``ScanNetXYZProbMultiDataset`` / ``ScanNetXYZProbSymDataset`` (utils/dataloader.py:89-210, :339-476, imported as they lie) over a miniature dataset written here
in the real on-disk formats: ScanNet ``scans/<id>/<id>_vh_clean_2.ply`` (binary little-endian, float xyz + uchar rgba
vertices, a face list after them), Scan2CAD ``full_annotations.json`` (scan trs + aligned models with trs / bbox /
center / catid_cad / sym), the per-model vertex-index ``segments`` pickle and the split text files.

The packages utils/dataloader.py imports that are absent from the image are bound to minimal stand-ins: ``plyfile``
(a reader for exactly the vertex layout written below), ``quaternion`` (np.quaternion / as_rotation_matrix: the
textbook unit-quaternion matrix, normalised like numpy-quaternion), ``MinkowskiEngine.utils.sparse_quantize``
(first point of every occupied voxel, in input order -- [ME-ext], the one convention here that the reference does not
pin: a different representative per voxel would permute/replace rows but not change the contract), ``h5py`` and ``hydra`` (unused by the class).

Writes tests/golden/scannet_mini/ (the dataset, 2 scans) and tests/golden/data_ref.npz (items for: no augmentation,
seeded augmentation with colour jitter, use_xyz features, and the category filters' scan lists).

    python tests/golden/make_data_golden.py            # needs /root/reference
"""
import json
import os
import pickle
import struct
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
MINI = os.path.join(HERE, "scannet_mini")
RES = 0.03


def mini_cfg(root=MINI, category="all", augment_color=False, use_xyz=False):
    data = types.SimpleNamespace(scan2cad=os.path.join(root, "full_annotations.json"), scannet=root,
                                 train_split=os.path.join(root, "train_split.txt"),
                                 val_split=os.path.join(root, "val_split.txt"),
                                 train_segments=os.path.join(root, "segments_train.pkl"),
                                 val_segments=os.path.join(root, "segments_val.pkl"))
    return types.SimpleNamespace(data=data, category=category, augment_color=augment_color, use_xyz=use_xyz,
                                 scannet_res=RES)


def _quat(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return [float(np.cos(angle / 2))] + [float(v) for v in np.sin(angle / 2) * axis]


def _qmat(q):
    w, x, y, z = q
    n = w * w + x * x + y * y + z * z
    s = 2.0 / n
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]])


def write_ply(path, xyz, rgb, faces):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    head = ("ply\nformat binary_little_endian 1.0\ncomment miniature ScanNet-format mesh (synthetic)\n"
            "element vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty uchar alpha\n"
            "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(xyz), len(faces)))
    v = np.zeros(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("a", "u1")])
    v["x"], v["y"], v["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    v["r"], v["g"], v["b"], v["a"] = rgb[:, 0], rgb[:, 1], rgb[:, 2], 255
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        f.write(v.tobytes())
        for tri in faces:
            f.write(struct.pack("<Biii", 3, *[int(i) for i in tri]))


def build_dataset(root=MINI):
    """Two scans.  World frame: y up.  Each aligned model is a box whose surface points are its segment."""
    rng = np.random.default_rng(2024)
    cats = ["03001627", "04379243", "99999999", "02871439", "03001627"]       # chair, table, unknown (-> 0), bookshelf, chair
    syms = ["__SYM_NONE", "__SYM_ROTATE_UP_2", "__SYM_NONE", "__SYM_ROTATE_UP_4", "__SYM_ROTATE_UP_INF"]
    annotations, segments = [], {}
    for si, id_scan in enumerate(["scene0000_00", "scene0001_00"]):
        scan_q = _quat([0.3, 1.0, -0.2], 0.7 + si)                            # scan -> world: a general rotation
        scan_t = [0.4 - si, 1.1, -0.6 + 0.5 * si]
        pts, seg, models = [], [], []
        floor = np.stack([rng.uniform(-2, 2, 700), rng.normal(0, 0.004, 700), rng.uniform(-2, 2, 700)], -1)
        pts.append(floor)
        base = len(floor)
        for mi in range(5 if si == 0 else 3):
            bbox = rng.uniform(0.2, 0.6, 3)                                   # CAD half extents in its own units
            sc = rng.uniform(0.7, 1.4, 3)
            if si == 0 and mi == 3:
                sc[1] = 5e-4                                                  # singular label: skipped (:171-172)
            center = rng.uniform(-0.05, 0.05, 3)
            q = _quat([0.05 * rng.normal(), 1.0, 0.05 * rng.normal()], rng.uniform(0, 2 * np.pi))
            if mi == 1:
                q = [1.0003 * v for v in q]                                   # annotations are unit only to ~1e-4
            t = [rng.uniform(-1.5, 1.5), 0.4, rng.uniform(-1.5, 1.5)]
            # surface samples of the unit cube, pushed through T R S T_center diag(bbox), plus duplicates per voxel
            m = 220
            u = rng.uniform(-1, 1, (m, 3))
            face = rng.integers(0, 3, m)
            u[np.arange(m), face] = rng.choice([-1.0, 1.0], m)
            local = (u * bbox + center) * sc
            world = local @ _qmat(q).T + np.asarray(t)
            world = np.concatenate([world, world[:40] + rng.normal(0, 0.002, (40, 3))])
            pts.append(world)
            seg.append(np.arange(base, base + len(world)))
            base += len(world)
            models.append({"catid_cad": cats[mi], "id_cad": "synthetic%02d" % mi, "sym": syms[mi],
                           "trs": {"translation": [float(v) for v in t], "rotation": [float(v) for v in q],
                                   "scale": [float(v) for v in sc]},
                           "bbox": [float(v) for v in bbox], "center": [float(v) for v in center]})
        world = np.concatenate(pts)
        perm = rng.permutation(len(world))                                    # vertex order is not grouped by object
        inv = np.empty_like(perm)
        inv[perm] = np.arange(len(perm))
        world = world[perm]
        seg = [np.sort(inv[s]) for s in seg]
        scan = (world - np.asarray(scan_t)) @ _qmat(scan_q)                   # inverse of T R (scale 1)
        rgb = rng.integers(0, 256, (len(scan), 3))
        faces = rng.integers(0, len(scan), (12, 3))
        write_ply(os.path.join(root, "scans", id_scan, id_scan + "_vh_clean_2.ply"), scan.astype(np.float32), rgb, faces)
        annotations.append({"id_scan": id_scan, "trs": {"translation": scan_t, "rotation": scan_q, "scale": [1.0, 1.0, 1.0]},
                            "aligned_models": models, "n_aligned_models": len(models)})
        segments[id_scan] = seg
    with open(os.path.join(root, "full_annotations.json"), "w") as f:
        json.dump(annotations, f)
    for name in ("train", "val"):
        with open(os.path.join(root, "segments_%s.pkl" % name), "wb") as f:
            pickle.dump(segments, f, protocol=2)
    with open(os.path.join(root, "val_split.txt"), "w") as f:
        f.write("scene0000_00\nscene0001_00\n")
    with open(os.path.join(root, "train_split.txt"), "w") as f:
        f.write("scene0001_00\n")


def standins():
    class _Ply(dict):
        @staticmethod
        def read(f):
            raw = f.read()
            end = raw.index(b"end_header\n") + len(b"end_header\n")
            n = int([l for l in raw[:end].decode().splitlines() if l.startswith("element vertex")][0].split()[2])
            v = np.frombuffer(raw, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"),
                                          ("blue", "u1"), ("alpha", "u1")], count=n, offset=end)
            return {"vertex": v}
    plyfile = types.ModuleType("plyfile")
    plyfile.PlyData = _Ply
    quaternion = types.ModuleType("quaternion")
    quaternion.as_rotation_matrix = lambda q: _qmat(q)
    np.quaternion = lambda w, x, y, z: (w, x, y, z)

    def sparse_quantize(coordinates, quantization_size=None, return_index=False, **_):
        c = np.floor(np.asarray(coordinates) / quantization_size).astype(np.int32)
        _, idx = np.unique(c, axis=0, return_index=True)
        idx = np.sort(idx)
        return (c[idx], idx) if return_index else c[idx]
    ME = types.ModuleType("MinkowskiEngine")
    ME.utils = types.SimpleNamespace(sparse_quantize=sparse_quantize)
    hydra = types.ModuleType("hydra")                                    # only decorates the module's own __main__ demo (:480-482)
    hydra.main = lambda **kw: (lambda fn: fn)
    return {"plyfile": plyfile, "quaternion": quaternion, "MinkowskiEngine": ME, "h5py": types.ModuleType("h5py"),
            "hydra": hydra}


if __name__ == "__main__":
    assert os.path.isdir("/root/reference/utils"), "run where /root/reference is mounted"
    build_dataset()
    sys.dont_write_bytecode = True
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.dataloader", "MinkowskiEngine")}
    sys.modules.update(standins())
    sys.path.insert(0, "/root/reference")
    try:
        from utils.dataloader import ScanNetXYZProbMultiDataset, ScanNetXYZProbSymDataset   # the reference's classes, as they lie
        out = {}

        def put(tag, item):
            out[tag + "_id"] = item[0]
            for name, a in zip(("coords", "feats", "xyz", "scale", "cls"), item[1:]):
                out[tag + "_" + name] = a

        ds = ScanNetXYZProbMultiDataset(mini_cfg(), training=False, augment=False)
        assert len(ds) == 2
        put("plain0", ds[0])
        put("plain1", ds[1])
        ds = ScanNetXYZProbMultiDataset(mini_cfg(augment_color=True), training=False, augment=True)
        np.random.seed(5)
        put("aug0", ds[0])
        put("aug1", ds[1])                                               # continues the same random stream
        ds = ScanNetXYZProbMultiDataset(mini_cfg(use_xyz=True), training=False, augment=True)
        np.random.seed(9)
        put("xyz1", ds[1])
        for cat in ("others", "03001627", "02871439", "04379243"):
            ds = ScanNetXYZProbMultiDataset(mini_cfg(category=cat), training=False, augment=False)
            out["scans_" + cat] = np.array([a["id_scan"] for a in ds.annotations])
            if cat in ("others", "02871439"):
                put("cat_" + cat, ds[0])
        # the symmetric dataset of train_separate.py (utils/dataloader.py:339-476)
        def put_sym(tag, item):
            out[tag + "_id"] = item[0]
            for name, a in zip(("coords", "feats", "scale", "obj", "cls"), (item[1], item[2], item[4], item[5], item[6])):
                out[tag + "_" + name] = a
            out[tag + "_nmodels"] = len(item[3])
            for mi, (rows, xyzs) in enumerate(item[3]):
                out["%s_m%d_rows" % (tag, mi)] = np.asarray(rows)
                out["%s_m%d_xyz" % (tag, mi)] = np.stack(xyzs).astype(np.float32)    # as collate_fn stores them

        ds = ScanNetXYZProbSymDataset(mini_cfg(), training=False, augment=False)
        put_sym("sym0", ds[0])
        ds = ScanNetXYZProbSymDataset(mini_cfg(category="03001627"), training=False, augment=True)
        np.random.seed(11)
        put_sym("symaug1", ds[1])
        np.savez_compressed(os.path.join(HERE, "data_ref.npz"), **out)
        print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.endswith(("coords", "_id")) or k.startswith("scans")})
    finally:
        sys.path.remove("/root/reference")
        for k in ("utils", "utils.dataloader", "MinkowskiEngine", "plyfile", "quaternion", "h5py", "hydra"):
            sys.modules.pop(k, None)
            if saved.get(k) is not None:
                sys.modules[k] = saved[k]
        if hasattr(np, "quaternion"):
            del np.quaternion
