"""Data contract of the step BEFORE the hot path: synthetic scenes and the real-file reader.

``SyntheticScanDataset[i]`` yields exactly what ``ScanNetXYZProbMultiDataset.__getitem__`` returns
(utils/dataloader.py:118-210): ``(id_scan, coords[N,3] float32 = floor(p/res), feats[N,3] rgb in [0,1],
xyz_labels[N,3], scale_labels[N,3], class_labels[N] int32 in 0..9)`` and ``collate_fn`` batches it like
train_joint.py:78-90 (batch index in column 0 of the int coordinates).  ``gt_lines`` gives the ground truth
in the text format eval_joint.py:285-301 reads (``tx ty tz ry sx sy sz ... category``).
There is no ScanNet/Scan2CAD data in this environment; ``ScanNetXYZProbMultiDataset`` below reads the real file
formats and is checked against the reference's own class on a miniature dataset (tests/golden/scannet_mini).
"""
import numpy as np
import torch

from .me import utils as me_utils
from .synth import make_scene


class SyntheticScanDataset(torch.utils.data.Dataset):
    def __init__(self, n_scenes=8, n_points=80000, res=0.03, seed0=0, **scene_kw):
        self.n_scenes, self.n_points, self.res, self.seed0, self.kw = n_scenes, n_points, res, seed0, scene_kw
        self._scenes = {}

    def __len__(self):
        return self.n_scenes

    def scene(self, index):
        if index not in self._scenes:
            self._scenes[index] = make_scene(self.seed0 + index, n_points=self.n_points, res=self.res, **self.kw)
        return self._scenes[index]

    def __getitem__(self, index):
        s = self.scene(index)
        return ("synth%04d" % (self.seed0 + index), s.coords.astype(np.float32), s.feats, s.xyz_labels,
                s.scale_labels, s.class_labels)

    def gt_lines(self, index):
        """eval_joint.py:287-288 line format: tx ty tz ry sx sy sz category"""
        return ["%f %f %f %f %f %f %f %d" % tuple(list(b[:7]) + [int(b[7])]) for b in self.scene(index).boxes]

    def gt(self, index):
        return parse_gt_lines(self.gt_lines(index))


def parse_gt_lines(lines):
    """eval_joint.py:285-301: ``tx ty tz ry sx sy sz ... category`` -> [(class index, (tx, ty, tz, ry, sx, sy, sz))].
    The category token is a ShapeNet catid or 'others' in the reference's results_gt files (mapped through
    idx2name, eval_joint.py:111-121) and a plain class index in the synthetic ones."""
    out = []
    for line in lines:
        tok = line.split(" ")
        if len(tok) < 8:
            continue
        cat = tok[-1].strip()
        idx = int(cat) if (cat.isdigit() and len(cat) < 8) else TOP8_CLASSES.get(cat, 0)
        out.append((idx, tuple(float(v) for v in tok[:7])))
    return out


def load_config(path, **overrides):
    """The reference's config/config.yaml layout (data.*, scannet_res, category, use_xyz, augment_color, ...) as
    attribute namespaces, without hydra; ``overrides`` replace top-level keys (``category='all'`` as
    eval_joint.py:137 / train_joint.py do)."""
    import types
    import yaml
    with open(path) as f:
        raw = yaml.safe_load(f)
    raw.pop("hydra", None)
    raw.update(overrides)

    def ns(d):
        return types.SimpleNamespace(**{k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})
    return ns(raw)


def collate_fn(batch):
    """train_joint.py:78-90"""
    id_scans, coords, feats, xyz_labels, scale_labels, class_labels = list(zip(*batch))
    coords_batch = me_utils.batched_coordinates(coords)
    return (id_scans, coords_batch, torch.from_numpy(np.concatenate(feats, 0)).float(),
            torch.from_numpy(np.concatenate(xyz_labels, 0)).float(),
            torch.from_numpy(np.concatenate(scale_labels, 0)).float(),
            torch.from_numpy(np.concatenate(class_labels, 0)).long())


# ----------------------------------------------------------------------------------------------------------------
# Real ScanNet + Scan2CAD files.  Same class name, constructor and item tuple as the reference's
# utils/dataloader.py:89-210 so train_joint.py / eval_joint.py-shaped callers take it unchanged; no plyfile /
# numpy-quaternion / MinkowskiEngine needed.  Checked against tests/golden/scannet_mini (the reference's own
# class run over a miniature dataset in the same on-disk formats, tests/golden/make_data_golden.py).

TOP8_CLASSES = {"03211117": 1, "04379243": 2, "02808440": 3, "02747177": 4, "04256520": 5, "03001627": 6,
                "02933112": 7, "02871439": 8}            # utils/dataloader.py:13-23; every other catid -> 0

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_vertices(path):
    """The ``vertex`` element of a PLY file as a numpy structured array (fields x, y, z, red, green, blue ... as the
    header names them).  ascii and binary (either endianness); the vertex element must come first, as it does in
    ScanNet's ``*_vh_clean_2.ply`` (what utils/dataloader.py:130-134 reads through plyfile)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, count, props, element = None, 0, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: PLY header has no end_header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if element is None and tok[1] != "vertex":
                    raise ValueError("%s: first PLY element is %r, expected vertex" % (path, tok[1]))
                element = tok[1]
                if element == "vertex":
                    count = int(tok[2])
            elif tok[0] == "property" and element == "vertex":
                if tok[1] == "list":
                    raise ValueError("%s: list property on the vertex element" % path)
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            out = np.zeros(count, dtype=[(n, t) for n, t in props])
            for i, (n, _) in enumerate(props):
                out[n] = rows[:, i]
            return out
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}.get(fmt)
        if order is None:
            raise ValueError("%s: unknown PLY format %r" % (path, fmt))
        dt = np.dtype([(n, order + t) for n, t in props])
        buf = f.read(dt.itemsize * count)
        if len(buf) != dt.itemsize * count:
            raise ValueError("%s: truncated vertex data" % path)
        return np.frombuffer(buf, dtype=dt, count=count)


def quat_matrix(q):
    """Rotation matrix of the quaternion (w, x, y, z); a non-unit quaternion rotates like its normalisation
    (numpy-quaternion's as_rotation_matrix, utils/dataloader.py:38,63,77)."""
    w, x, y, z = (float(v) for v in q)
    s = 2.0 / (w * w + x * x + y * y + z * z)
    return np.array([[1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
                     [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
                     [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)]], np.float64)


def _affine(linear=None, shift=None):
    M = np.eye(4)
    if linear is not None:
        M[:3, :3] = linear
    if shift is not None:
        M[:3, 3] = shift
    return M


def trs_matrix(t, q, s):
    """T R S (utils/dataloader.py:72-82)"""
    return _affine(shift=np.asarray(t, np.float64)) @ _affine(quat_matrix(q)) @ _affine(np.diag(np.asarray(s, np.float64)))


def bbox_matrix(model):
    """unit cube -> world: T R S T_center diag(bbox) of one Scan2CAD aligned model (utils/dataloader.py:49-69)"""
    trs = model["trs"]
    return (trs_matrix(trs["translation"], trs["rotation"], trs["scale"])
            @ _affine(shift=np.asarray(model["center"], np.float64)) @ _affine(np.diag(np.asarray(model["bbox"], np.float64))))


def transform_points(pc, M):
    """utils/dataloader.py:85-86 (homogeneous product in float64)"""
    return (M @ np.concatenate([pc, np.ones((pc.shape[0], 1))], -1).T).T[:, :3]


class ScanNetXYZProbMultiDataset(torch.utils.data.Dataset):
    """utils/dataloader.py:89-210.  ``cfg`` needs ``data.scan2cad`` (full_annotations.json), ``data.scannet`` (root with
    scans/<id>/<id>_vh_clean_2.ply), ``data.{train,val}_split`` (text, one scan id per line),
    ``data.{train,val}_segments`` (pickle: scan id -> per aligned model, vertex indices), ``category``
    ('all' | 'others' | a catid), ``augment_color``, ``use_xyz``, ``scannet_res``.  Augmentation draws from numpy's
    global generator in the reference's order, so a seeded run reproduces the reference's samples."""

    def __init__(self, cfg, training, augment):
        import json
        import pickle
        self.cfg, self.training, self.augment = cfg, training, augment
        with open(cfg.data.scan2cad) as f:
            annotations = json.load(f)
        with open(cfg.data.train_split if training else cfg.data.val_split) as f:
            wanted = set(f.read().splitlines())
        with open(cfg.data.train_segments if training else cfg.data.val_segments, "rb") as f:
            self.segments = pickle.load(f)
        # :102-110: with a category filter, scans without a matching model are dropped; 'all' keeps every scan
        self.annotations = [a for a in annotations if a["id_scan"] in wanted
                            and (cfg.category == "all" or self._select(a["aligned_models"]))]

    def class_index(self, catid):
        return TOP8_CLASSES.get(catid, 0)

    def _select(self, models):
        cat = self.cfg.category
        if cat == "all":
            return list(models)
        if cat == "others":
            return [m for m in models if self.class_index(m["catid_cad"]) == 0]
        return [m for m in models if m["catid_cad"] == cat]

    def __len__(self):
        return len(self.annotations)

    def gt(self, index):
        """ground-truth boxes of scan ``index`` from ``cfg.data.gt_path/<id_scan>.txt`` (eval_joint.py:285)"""
        import os
        with open(os.path.join(self.cfg.data.gt_path, self.annotations[index]["id_scan"] + ".txt")) as f:
            return parse_gt_lines(f.read().splitlines())

    def __getitem__(self, index):
        import os
        ann = self.annotations[index]
        id_scan = ann["id_scan"]
        if not np.all(np.abs(np.asarray(ann["trs"]["scale"]) - 1.0) < 1e-7):
            raise ValueError("%s: scan transform has a non-unit scale" % id_scan)
        path = os.path.join(self.cfg.data.scannet, "scans", id_scan, id_scan + "_vh_clean_2.ply")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " does not exist.")
        v = read_ply_vertices(path)
        rgb = (np.stack([v["red"], v["green"], v["blue"]], -1) / 255.0).astype(np.float32)
        to_world = trs_matrix(ann["trs"]["translation"], ann["trs"]["rotation"], ann["trs"]["scale"])
        points = transform_points(np.stack([v["x"], v["y"], v["z"]], -1), to_world)
        for model, seg in zip(ann["aligned_models"], self.segments[id_scan]):
            model["segments"] = seg
        models = self._select(ann["aligned_models"])
        if not models:
            return self[np.random.randint(len(self))]
        aug = np.eye(4)
        if self.augment:
            if self.cfg.augment_color:                                   # :156-160, float32 in place
                rgb *= (1 + 0.4 * np.random.random(3) - 0.2)
                rgb += (0.1 * np.random.random(3) - 0.05)
                rgb += (0.05 * np.random.random(points.shape[0]) - 0.025)[:, None]
                rgb = np.clip(rgb, 0, 1)
            angle = np.random.randint(4) * np.pi / 2.0 + (np.random.random() - 0.5) * 2.0 * np.pi / 9.0
            c, s = np.cos(angle), np.sin(angle)
            Ry = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
            points = points @ Ry.T
            aug[:3, :3] = Ry
        points = points.astype(np.float32)
        n = points.shape[0]
        xyz = np.zeros((n, 3), np.float32)
        scale = np.zeros((n, 3), np.float32)
        cls = np.full(n, 9, np.int32)
        for m in models:
            s32 = np.asarray(m["trs"]["scale"], np.float32)
            if s32.min() < 1e-3:                                         # singular annotation (:171-172)
                continue
            seg = m["segments"]
            xyz[seg] = transform_points(points[seg], np.linalg.inv(aug @ bbox_matrix(m)))
            scale[seg] = s32 * np.asarray(m["bbox"], np.float32)         # half extents in metres
            cls[seg] = self.class_index(m["catid_cad"])
        feats = np.concatenate([points, rgb], -1) if self.cfg.use_xyz else rgb
        keep = me_utils.sparse_quantize(np.ascontiguousarray(points), quantization_size=self.cfg.scannet_res,
                                        return_index=True)[1]
        coords = np.floor(points[keep] / self.cfg.scannet_res).astype(np.float32)
        return id_scan, coords, feats[keep], xyz[keep], scale[keep], cls[keep]


def _roty4(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]])


# rotations about the up axis under which a CAD model looks the same (utils/dataloader.py:444-453)
SYMMETRY_ANGLES = {"__SYM_ROTATE_UP_2": [np.pi], "__SYM_ROTATE_UP_4": [np.pi / 2, np.pi, -np.pi / 2],
                   "__SYM_ROTATE_UP_INF": [2 * np.pi / 36 * i for i in range(1, 36)]}


class ScanNetXYZProbSymDataset(ScanNetXYZProbMultiDataset):
    """utils/dataloader.py:339-476, the dataset of train_separate.py: same files, but the scan is quantised FIRST and
    the object labels are kept per aligned model as ``[rows of the model's points, [their LCC coordinates under each
    symmetry-equivalent pose]]`` so the loss can take the best pose; item = (id_scan, coords, feats, xyz_labels,
    scale_labels, obj_labels (0/1), class_labels (0 = background or 'others'))."""

    def __getitem__(self, index):
        import os
        ann = self.annotations[index]
        id_scan = ann["id_scan"]
        if not np.all(np.abs(np.asarray(ann["trs"]["scale"]) - 1.0) < 1e-7):
            raise ValueError("%s: scan transform has a non-unit scale" % id_scan)
        path = os.path.join(self.cfg.data.scannet, "scans", id_scan, id_scan + "_vh_clean_2.ply")
        if not os.path.exists(path):
            raise FileNotFoundError(path + " does not exist.")
        v = read_ply_vertices(path)
        rgb = np.stack([v["red"], v["green"], v["blue"]], -1)               # stays uint8 until after quantisation (:383)
        to_world = trs_matrix(ann["trs"]["translation"], ann["trs"]["rotation"], ann["trs"]["scale"])
        points = transform_points(np.stack([v["x"], v["y"], v["z"]], -1), to_world)
        for model, seg in zip(ann["aligned_models"], self.segments[id_scan]):
            model["segments"] = seg
        models = self._select(ann["aligned_models"])
        if not models:
            return self[np.random.randint(len(self))]
        aug = np.eye(4)
        if self.augment:
            if self.cfg.augment_color:                                   # :400-404 (on the raw 0..255 colours)
                rgb = rgb.astype(np.float64)
                rgb *= (1 + 0.4 * np.random.random(3) - 0.2)
                rgb += (0.1 * np.random.random(3) - 0.05)
                rgb += (0.05 * np.random.random(points.shape[0]) - 0.025)[:, None]
                rgb = np.clip(rgb, 0, 1)
            angle = np.random.randint(4) * np.pi / 2.0 + (np.random.random() - 0.5) * 2.0 * np.pi / 9.0
            c, s = np.cos(angle), np.sin(angle)
            Ry = np.array([[c, 0, -s], [0, 1, 0], [s, 0, c]])
            points = points @ Ry.T
            aug[:3, :3] = Ry
        points = points.astype(np.float32)
        keep = me_utils.sparse_quantize(np.ascontiguousarray(points), quantization_size=self.cfg.scannet_res,
                                        return_index=True)[1]
        points = points[keep]
        rgb = (rgb[keep] / 255.0).astype(np.float32)
        new_row = np.full(v.shape[0], -1, np.int64)
        new_row[keep] = np.arange(len(keep))
        coords = np.floor(points / self.cfg.scannet_res).astype(np.float32)
        n = points.shape[0]
        scale = np.zeros((n, 3), np.float32)
        obj = np.zeros(n, np.int32)
        cls = np.zeros(n, np.int32)
        xyz_labels = []
        for m in models:
            s32 = np.asarray(m["trs"]["scale"], np.float32)
            if s32.min() < 1e-3:
                continue
            M = bbox_matrix(m)
            poses = [M] + [M @ _roty4(a) for a in SYMMETRY_ANGLES.get(m["sym"], [])]
            rows = new_row[np.asarray(m["segments"], np.int64)]
            rows = rows[rows >= 0]                                       # the model's points that survived quantisation
            pts = points[rows]
            xyzs = [transform_points(pts, np.linalg.inv(aug @ P)) for P in poses]
            scale[rows] = s32 * np.asarray(m["bbox"], np.float32)
            obj[rows] = 1
            cls[rows] = self.class_index(m["catid_cad"])
            xyz_labels.append([rows, xyzs])
        feats = np.concatenate([points, rgb], -1) if self.cfg.use_xyz else rgb
        return id_scan, coords, feats, xyz_labels, scale, obj, cls


def collate_fn_separate(batch):
    """train_separate.py:78-97: like collate_fn, the per-model label lists stay per scan"""
    id_scans, coords, feats, xyz_labels, scale_labels, obj_labels, class_labels = list(zip(*batch))
    xyz_batch = [[[torch.from_numpy(np.asarray(rows)).long(), [torch.from_numpy(x).float() for x in xyzs]]
                  for rows, xyzs in per_scan] for per_scan in xyz_labels]
    return (id_scans, me_utils.batched_coordinates(coords), torch.from_numpy(np.concatenate(feats, 0)).float(), xyz_batch,
            torch.from_numpy(np.concatenate(scale_labels, 0)).float(),
            torch.from_numpy(np.concatenate(obj_labels, 0)).long(),
            torch.from_numpy(np.concatenate(class_labels, 0)).long())
