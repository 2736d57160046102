"""Data contract of the step BEFORE the hot path on synthetic scenes.

``SyntheticScanDataset[i]`` yields exactly what ``ScanNetXYZProbMultiDataset.__getitem__`` returns
(utils/dataloader.py:118-210): ``(id_scan, coords[N,3] float32 = floor(p/res), feats[N,3] rgb in [0,1],
xyz_labels[N,3], scale_labels[N,3], class_labels[N] int32 in 0..9)`` and ``collate_fn`` batches it like
train_joint.py:78-90 (batch index in column 0 of the int coordinates).  ``gt_lines`` gives the ground truth
in the text format eval_joint.py:285-301 reads (``tx ty tz ry sx sy sz ... category``).
There is no ScanNet/Scan2CAD data in this environment; a reader for the real files is future work.
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


def collate_fn(batch):
    """train_joint.py:78-90"""
    id_scans, coords, feats, xyz_labels, scale_labels, class_labels = list(zip(*batch))
    coords_batch = me_utils.batched_coordinates(coords)
    return (id_scans, coords_batch, torch.from_numpy(np.concatenate(feats, 0)).float(),
            torch.from_numpy(np.concatenate(xyz_labels, 0)).float(),
            torch.from_numpy(np.concatenate(scale_labels, 0)).float(),
            torch.from_numpy(np.concatenate(class_labels, 0)).long())
