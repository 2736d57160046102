"""Golden vectors for the joint training loss, produced by executing the reference's own lines
(train_joint.py:253-283: head gather by ground-truth class, masked MSE(xyz), MSE(log scale), CE(class)) on CPU torch
tensors in the build container; `.cuda()` patched to a plain copy (it matters: the reference zeroes class 9 through an expanded view of the
device copy of the labels, :254-255, and computes the object mask from the untouched host labels, :260).  Stores the three loss terms, their sum and the gradient
of the sum w.r.t. the network output (tests/golden/loss_ref.npz); tests/test_oracle_decode.py compares
canonicalvoting_amd.train.joint_loss with them.  Same for the per-category model's loss with the minimum over
symmetry-equivalent poses (train_separate.py:247-286 -> the ``sep_*`` entries, canonicalvoting_amd.train.separate_loss).

    python tests/golden/make_loss_golden.py            # needs /root/reference
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/train_joint.py"
REF_SEP = "/root/reference/train_separate.py"


def ref_lines(a, b, path=None):
    src = open(path or REF).read().splitlines()
    return textwrap.dedent("\n".join(src[a - 1:b])) + "\n"


def make_inputs(seed=7, n=4000, nclasses=9):
    rng = np.random.default_rng(seed)
    F = rng.normal(0, 0.5, (n, 7 * nclasses + 1)).astype(np.float32)
    labels = rng.integers(-1, nclasses + 1, n).astype(np.int64)           # -1 and 9 are both "no object"
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    scale = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    return F, labels, xyz, scale


def make_separate_inputs(seed=13):
    """a batch of the two scans of tests/golden/scannet_mini through the product's symmetric dataset + collate
    (both bit-equal to the reference's, tests/test_data.py) and a seeded 8-channel network output"""
    from canonicalvoting_amd import data
    from tests.golden.make_data_golden import mini_cfg
    ds = data.ScanNetXYZProbSymDataset(mini_cfg(), training=False, augment=False)
    batch = data.collate_fn_separate([ds[0], ds[1]])
    rng = np.random.default_rng(seed)
    F = rng.normal(0, 0.5, (batch[1].shape[0], 8)).astype(np.float32)
    return batch, F


def separate_golden():
    """train_separate.py:237-238 (unpack, mask) and :247-286 (head split, three losses, sum) executed as they lie"""
    batch, F = make_separate_inputs()
    out_F = torch.from_numpy(F.copy()).requires_grad_(True)
    ns = {"torch": torch, "cfg": types.SimpleNamespace(log_scale=True, xyz_factor=1.0, scale_factor=1.0, batch_size=2),
          "data": batch, "scan_output": types.SimpleNamespace(F=out_F), "xyz_weights": torch.tensor([1.0, 1.0, 1.0]),
          "obj_criterion": torch.nn.CrossEntropyLoss(), "losses": {}}
    exec(ref_lines(237, 238, REF_SEP), ns)
    exec(ref_lines(247, 286, REF_SEP), ns)
    ns["loss"].backward()
    return {"sep_loss": float(ns["loss"]), "sep_loss_obj": float(ns["loss_obj"]), "sep_loss_xyz": float(ns["loss_xyz"]),
            "sep_loss_scale": float(ns["loss_scale"]), "sep_grad": out_F.grad.numpy().astype(np.float32)}


if __name__ == "__main__":
    assert os.path.exists(REF), "run where /root/reference is mounted"
    F, labels, xyz, scale = make_inputs()
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()      # a host-to-device transfer is a copy
    try:
        out_F = torch.from_numpy(F.copy()).requires_grad_(True)
        ns = {"torch": torch, "nclasses": 9, "cfg": types.SimpleNamespace(log_scale=True, xyz_factor=1.0, scale_factor=1.0),
              "scan_output": types.SimpleNamespace(F=out_F), "scan_class_labels": torch.from_numpy(labels.copy()),
              "scan_xyz_labels": torch.from_numpy(xyz), "scan_scale_labels": torch.from_numpy(scale),
              "xyz_weights": torch.tensor([1.0, 1.0, 1.0]), "obj_criterion": torch.nn.CrossEntropyLoss(), "losses": {}}
        # the reference feeds labels -1 (unlabelled) to CrossEntropyLoss only after its dataloader mapped them to 9
        # (utils/dataloader.py:172); keep the raw labels for the mask and clamp the CE target the same way
        ns["scan_class_labels"] = torch.from_numpy(np.where(labels < 0, 9, labels))
        exec(ref_lines(253, 283), ns)                     # ends with the reference's own loss.backward()
        sep = separate_golden()
        print("separate", {k: v for k, v in sep.items() if k != "sep_grad"})
        np.savez_compressed(os.path.join(HERE, "loss_ref.npz"), seed=7, n=4000, **sep,
                            loss=float(ns["loss"]), loss_xyz=float(ns["loss_xyz"]), loss_scale=float(ns["loss_scale"]),
                            loss_class=float(ns["loss_class"]), grad=out_F.grad.numpy().astype(np.float32))
        print("loss", float(ns["loss"]), float(ns["loss_xyz"]), float(ns["loss_scale"]), float(ns["loss_class"]))
    finally:
        torch.Tensor.cuda = orig_cuda
