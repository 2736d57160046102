"""Golden vectors for the joint training loss, produced by executing the reference's own lines
(train_joint.py:253-283: head gather by ground-truth class, masked MSE(xyz), MSE(log scale), CE(class)) on CPU torch
tensors in the build container; `.cuda()` patched to a plain copy (it matters: the reference zeroes class 9 through an expanded view of the
device copy of the labels, :254-255, and computes the object mask from the untouched host labels, :260).  Stores the three loss terms, their sum and the gradient
of the sum w.r.t. the network output (tests/golden/loss_ref.npz); tests/test_oracle_decode.py compares
canonicalvoting_amd.train.joint_loss with them.

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


def ref_lines(a, b):
    src = open(REF).read().splitlines()
    return textwrap.dedent("\n".join(src[a - 1:b])) + "\n"


def make_inputs(seed=7, n=4000, nclasses=9):
    rng = np.random.default_rng(seed)
    F = rng.normal(0, 0.5, (n, 7 * nclasses + 1)).astype(np.float32)
    labels = rng.integers(-1, nclasses + 1, n).astype(np.int64)           # -1 and 9 are both "no object"
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    scale = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    return F, labels, xyz, scale


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
        np.savez_compressed(os.path.join(HERE, "loss_ref.npz"), seed=7, n=4000,
                            loss=float(ns["loss"]), loss_xyz=float(ns["loss_xyz"]), loss_scale=float(ns["loss_scale"]),
                            loss_class=float(ns["loss_class"]), grad=out_F.grad.numpy().astype(np.float32))
        print("loss", float(ns["loss"]), float(ns["loss_xyz"]), float(ns["loss_scale"]), float(ns["loss_class"]))
    finally:
        torch.Tensor.cuda = orig_cuda
