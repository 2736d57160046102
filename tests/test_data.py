"""Data contract: the real-file reader against goldens the reference's own dataset class produced
(tests/golden/make_data_golden.py -> scannet_mini/ + data_ref.npz), the PLY reader, and the synthetic dataset."""
import os
import struct

import numpy as np
import pytest

from canonicalvoting_amd import data
from tests.golden.make_data_golden import MINI, mini_cfg

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_ref.npz"))


def check(tag, item, exact=True):
    assert item[0] == str(GOLD[tag + "_id"])
    for name, a in zip(("coords", "feats", "xyz", "scale", "cls"), item[1:]):
        g = GOLD[tag + "_" + name]
        assert a.dtype == g.dtype and a.shape == g.shape, (tag, name, a.dtype, g.dtype, a.shape, g.shape)
        if exact:
            assert np.array_equal(a, g), (tag, name, np.abs(a.astype(np.float64) - g).max())
        else:
            np.testing.assert_allclose(a, g, rtol=0, atol=1e-6)


def test_reader_matches_reference_class_run_on_the_mini_dataset():
    ds = data.ScanNetXYZProbMultiDataset(mini_cfg(), training=False, augment=False)
    assert len(ds) == 2
    check("plain0", ds[0])
    check("plain1", ds[1])
    # labels: background 9, unknown catid 0, the singular-scale model skipped
    cls = ds[0][5]
    assert set(np.unique(cls)) == {0, 2, 6, 9}
    assert len(data.ScanNetXYZProbMultiDataset(mini_cfg(), training=True, augment=False)) == 1


def test_seeded_augmentation_reproduces_the_reference_samples():
    ds = data.ScanNetXYZProbMultiDataset(mini_cfg(augment_color=True), training=False, augment=True)
    np.random.seed(5)
    check("aug0", ds[0])
    check("aug1", ds[1])
    ds = data.ScanNetXYZProbMultiDataset(mini_cfg(use_xyz=True), training=False, augment=True)
    np.random.seed(9)
    item = ds[1]
    check("xyz1", item)
    assert item[2].shape[1] == 6


@pytest.mark.parametrize("cat", ["others", "03001627", "02871439", "04379243"])
def test_category_filters(cat):
    ds = data.ScanNetXYZProbMultiDataset(mini_cfg(category=cat), training=False, augment=False)
    assert [a["id_scan"] for a in ds.annotations] == [str(s) for s in GOLD["scans_" + cat]]
    if cat in ("others", "02871439"):
        check("cat_" + cat, ds[0])
    if cat == "02871439":
        assert (ds[0][5] == 9).all()          # its only bookshelf is the singular one: every point stays background


def test_quaternion_matrix_against_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4)                                           # (w, x, y, z), not normalised
        want = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()  # scipy: (x, y, z, w), normalises
        np.testing.assert_allclose(data.quat_matrix(q), want, atol=1e-12)


def test_collate_of_real_items_and_label_geometry():
    ds = data.ScanNetXYZProbMultiDataset(mini_cfg(), training=False, augment=False)
    ids, coords, feats, xyz, scale, cls = data.collate_fn([ds[0], ds[1]])
    assert coords.shape[1] == 4 and coords[:, 0].unique().tolist() == [0, 1] and coords.dtype.is_floating_point is False
    assert feats.shape[0] == coords.shape[0] == cls.shape[0]
    obj = (cls != 9)
    # an object's points lie on its unit-cube surface in label space; half extents are positive
    assert float(xyz[obj].abs().max(1).values.sub(1).abs().max()) < 0.05
    assert (scale[obj] > 0).all() and (scale[~obj] == 0).all()


def _tiny_ply(tmp_path, fmt):
    xyz = np.array([[0.5, -1.25, 2.0], [3.0, 4.5, -6.0], [7.0, 8.0, 9.0]], np.float32)
    rgb = np.array([[1, 2, 3], [40, 50, 60], [255, 0, 128]], np.uint8)
    head = ("ply\nformat %s 1.0\ncomment x\nobj_info y\nelement vertex 3\nproperty float x\nproperty float y\n"
            "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty double q\n"
            "element face 1\nproperty list uchar int vertex_indices\nend_header\n" % fmt)
    p = tmp_path / (fmt + ".ply")
    with open(p, "wb") as f:
        f.write(head.encode())
        for i in range(3):
            if fmt == "ascii":
                f.write(("%r %r %r %d %d %d %r\n" % (*[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]], 0.25 * i)).encode())
            else:
                e = "<" if fmt == "binary_little_endian" else ">"
                f.write(struct.pack(e + "fffBBBd", *xyz[i], *rgb[i], 0.25 * i))
        f.write(b"3 0 1 2\n" if fmt == "ascii" else struct.pack(("<" if "little" in fmt else ">") + "Biii", 3, 0, 1, 2))
    return str(p), xyz, rgb


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_reader_formats(tmp_path, fmt):
    path, xyz, rgb = _tiny_ply(tmp_path, fmt)
    v = data.read_ply_vertices(path)
    assert len(v) == 3
    assert np.array_equal(np.stack([v["x"], v["y"], v["z"]], -1), xyz)
    assert np.array_equal(np.stack([v["red"], v["green"], v["blue"]], -1), rgb)
    assert np.array_equal(v["q"], [0.0, 0.25, 0.5])


def test_ply_reader_rejects_bad_files(tmp_path):
    p = tmp_path / "a.ply"
    p.write_bytes(b"plx\n")
    with pytest.raises(ValueError, match="not a PLY"):
        data.read_ply_vertices(str(p))
    p.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nend_header\n\0\0\0\0")
    with pytest.raises(ValueError, match="truncated"):
        data.read_ply_vertices(str(p))
    p.write_bytes(b"ply\nformat ascii 1.0\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n")
    with pytest.raises(ValueError, match="expected vertex"):
        data.read_ply_vertices(str(p))
    cfg = mini_cfg()
    ds = data.ScanNetXYZProbMultiDataset(cfg, training=False, augment=False)
    cfg.data.scannet = str(tmp_path)
    with pytest.raises(FileNotFoundError, match="does not exist"):
        ds[0]


def test_synthetic_dataset_contract():
    ds = data.SyntheticScanDataset(n_scenes=2, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5,
                                   box_scale=0.4)
    item = ds[1]
    assert item[1].dtype == np.float32 and item[1].shape == (1500, 3) and item[5].dtype == np.int32
    assert len(ds.gt_lines(1)) == 2 and len(ds.gt_lines(1)[0].split()) == 8


def test_config_yaml_and_gt_files(tmp_path):
    (tmp_path / "results_gt").mkdir()
    (tmp_path / "results_gt" / "scene0000_00.txt").write_text(
        "0.5 0.4 -1.0 0.3 0.4 0.5 0.6 03001627\n1 2 3 0.1 0.2 0.3 0.4 0.9 others\n-1 0 1 0 1 1 1 04379243\n\n")
    (tmp_path / "config.yaml").write_text(
        "data:\n    scan2cad: %s/full_annotations.json\n    scannet: %s\n    train_split: %s/train_split.txt\n"
        "    val_split: %s/val_split.txt\n    train_segments: %s/segments_train.pkl\n    val_segments: %s/segments_val.pkl\n"
        "    gt_path: %s/results_gt\nscannet_res: 0.03\nnum_workers: 0\nbatch_size: 3\naugment_color: False\naugment: True\n"
        "use_xyz: False\ncategory: !!str '03001627'\nopt:\n    learning_rate: 1e-3\nhydra:\n    run:\n        dir: x\n"
        % ((MINI,) * 6 + (str(tmp_path),)))
    cfg = data.load_config(str(tmp_path / "config.yaml"), category="all")
    assert cfg.category == "all" and cfg.scannet_res == 0.03 and not hasattr(cfg, "hydra") and cfg.data.scannet == MINI
    ds = data.ScanNetXYZProbMultiDataset(cfg, training=False, augment=False)
    check("plain0", ds[0])
    gt = ds.gt(0)
    assert [c for c, _ in gt] == [6, 0, 2]
    assert gt[0][1] == (0.5, 0.4, -1.0, 0.3, 0.4, 0.5, 0.6) and gt[1][1][:3] == (1.0, 2.0, 3.0)
    syn = data.SyntheticScanDataset(n_scenes=1, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5,
                                    box_scale=0.4)
    assert [c for c, _ in syn.gt(0)] == [int(b[7]) for b in syn.scene(0).boxes]


def check_sym(tag, item):
    assert item[0] == str(GOLD[tag + "_id"])
    for name, a in zip(("coords", "feats", "scale", "obj", "cls"), (item[1], item[2], item[4], item[5], item[6])):
        g = GOLD[tag + "_" + name]
        assert a.dtype == g.dtype and np.array_equal(a, g), (tag, name)
    assert len(item[3]) == int(GOLD[tag + "_nmodels"])
    for mi, (rows, xyzs) in enumerate(item[3]):
        assert np.array_equal(rows, GOLD["%s_m%d_rows" % (tag, mi)])
        assert np.array_equal(np.stack(xyzs).astype(np.float32), GOLD["%s_m%d_xyz" % (tag, mi)])


def test_symmetric_dataset_matches_reference_class():
    ds = data.ScanNetXYZProbSymDataset(mini_cfg(), training=False, augment=False)
    item = ds[0]
    check_sym("sym0", item)
    # poses per model: none 1, UP_2 2, (unknown-category model) 1, [singular one skipped], UP_INF 36
    assert [len(x) for _, x in item[3]] == [1, 2, 1, 36]
    ds = data.ScanNetXYZProbSymDataset(mini_cfg(category="03001627"), training=False, augment=True)
    np.random.seed(11)
    check_sym("symaug1", ds[1])
    batch = data.collate_fn_separate([item, item])
    assert batch[1].shape[1] == 4 and len(batch[3]) == 2 and batch[3][0][1][1][0].dtype.is_floating_point
    assert batch[5].dtype == batch[6].dtype and int(batch[5].max()) == 1
