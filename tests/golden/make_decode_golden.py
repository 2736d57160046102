"""Golden vectors for the head split and the greedy decode loop, produced by EXECUTING THE REFERENCE'S OWN LINES.

The decode loop and the head split of qq456cvb/CanonicalVoting are inline in eval_joint.py:main() (they cannot be
imported; the module's top-level imports - hydra, MinkowskiEngine, hv_cuda - are absent here).  This script, run in
the build container where /root/reference is mounted, reads eval_joint.py, slices out

    :18-21    thresh_high / thresh_low / valid_ratio / elimination
    :41-46    unravel_index
    :173-190  head split  (scan_output.F -> xyz_pred, scale_pred, class_pred, prob_pred)
    :195-263  the greedy decode loop (argmax -> box -> suppression -> back-projection check -> class / score)

and exec()s them unchanged on CPU torch tensors (`.cuda()` / `.to('cuda')` patched to a plain copy; the vote grids the
loop consumes come from oracle/hv_oracle.c, because hv_cuda itself only exists as CUDA source).  Nothing of the
reference's text is stored: the fixture holds inputs and the outputs those lines produced
(tests/golden/decode_ref_*.npz).  tests/test_oracle_decode.py checks oracle/decode_oracle.c and
oracle/sparse_oracle.head_joint_eval against them on the CPU; tests/test_decode_gpu.py checks the HIP decode.

    python tests/golden/make_decode_golden.py            # needs /root/reference
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/eval_joint.py"


def ref_lines(a, b):
    src = open(REF).read().splitlines()
    return textwrap.dedent("\n".join(src[a - 1:b])) + "\n"


def run_reference(scan_output_F, coords, res, nclasses=9):
    """executes eval_joint.py:173-190 and :195-263 on CPU tensors; returns a dict of inputs and outputs"""
    import oracle
    # identity .cuda() / .to('cuda') for the duration of the run
    orig_cuda, orig_to = torch.Tensor.cuda, torch.Tensor.to
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()      # a host-to-device transfer is a copy

    def to(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
        k = {kk: v for kk, v in k.items() if not (kk == "device" and str(v).startswith("cuda"))}
        return orig_to(self, *a, **k) if (a or k) else self.clone()
    torch.Tensor.to = to
    orig_tensor = torch.tensor
    try:
        ns = {"torch": torch, "np": np, "nclasses": nclasses, "SCENENN": False,
              "cfg": types.SimpleNamespace(scannet_res=res, log_scale=True, use_xyz=False)}
        exec(ref_lines(18, 21), ns)                                  # thresholds
        exec(ref_lines(41, 46), ns)                                  # unravel_index
        ns["scan_output"] = types.SimpleNamespace(F=torch.from_numpy(scan_output_F.copy()))
        ns["scan_points"] = torch.from_numpy(np.concatenate([np.zeros((len(coords), 1), np.int64), coords], 1))
        exec(ref_lines(173, 190), ns)                                # head split
        head = {k: ns[k].detach().numpy().copy() for k in ("xyz_pred", "scale_pred", "class_pred", "prob_pred")}
        pts = (coords * np.float32(res)).astype(np.float32)          # :193 curr_points * res in float32
        assert np.array_equal(pts, (ns["curr_points"].float() * res).numpy())
        g = oracle.hv_forward(pts, head["xyz_pred"], head["scale_pred"], head["prob_pred"], res, 120)
        ns["grid_obj"], ns["grid_rot"], ns["grid_scale"] = [torch.from_numpy(a.copy()) for a in g[:3]]
        ns["curr_points"] = ns["curr_points"].float()
        exec(ref_lines(195, 263), ns)                                # decode loop (leaves boxes / scores / classes)
        out = dict(boxes=np.array(ns["boxes"], np.float32).reshape(-1, 8, 3), scores=np.array(ns["scores"], np.float32),
                   classes=np.array(ns["classes"], np.int64), grid_obj_after=ns["grid_obj"].numpy().copy(),
                   thresh_high=ns["thresh_high"], thresh_low=ns["thresh_low"], valid_ratio=ns["valid_ratio"],
                   elimination=ns["elimination"])
        return head, g, out
    finally:
        torch.Tensor.cuda, torch.Tensor.to, torch.tensor = orig_cuda, orig_to, orig_tensor


def make_case(seed, n_points, thresh_scale):
    """scan_output whose head split gives usable predictions: label-derived xyz / log-scale in the predicted class's
    slot, class logits peaked at the label (background rows peaked at class 9), some noise."""
    from canonicalvoting_amd.synth import make_scene
    sc = make_scene(seed, n_points=n_points)
    rng = np.random.default_rng(seed)
    n, nc = n_points, 9
    F = rng.normal(0, 0.05, (n, 7 * nc + 1)).astype(np.float32)
    lab = sc.class_labels.astype(np.int64)
    obj = lab < nc
    rows = np.nonzero(obj)[0]
    for d in range(3):
        F[rows, lab[rows] * 3 + d] = sc.xyz_labels[rows, d] + rng.normal(0, 0.03, len(rows))
        F[rows, 3 * nc + lab[rows] * 3 + d] = np.log(sc.scale_labels[rows, d]) + rng.normal(0, 0.03, len(rows))
    F[np.arange(n), 6 * nc + np.where(obj, lab, nc)] += 6.0
    return sc, F


CASES = (("decode_ref_8k", 3, 8000), ("decode_ref_5k", 5, 5000))


if __name__ == "__main__":
    assert os.path.exists(REF), "run where /root/reference is mounted"
    for name, seed, n in CASES:
        sc, F = make_case(seed, n, 1.0)
        head, g, out = run_reference(F, sc.coords.astype(np.int64), sc.res)
        print(name, "grid", g[0].shape, "max obj %.1f" % g[0].max(), "boxes", len(out["boxes"]), "classes", out["classes"].tolist())
        # the network output is regenerated from (seed, n) by make_case; the fixture keeps what the reference's lines
        # produced from it
        np.savez_compressed(os.path.join(HERE, name + ".npz"), seed=seed, n=n, res=np.float32(sc.res),
                            xyz_pred=head["xyz_pred"], scale_pred=head["scale_pred"],
                            class_pred=head["class_pred"].astype(np.int16), prob_pred=head["prob_pred"], boxes=out["boxes"],
                            scores=out["scores"], classes=out["classes"],
                            zeroed=np.flatnonzero((g[0] != 0) & (out["grid_obj_after"] == 0)).astype(np.int32),
                            consts=np.array([out["thresh_high"], out["thresh_low"], out["valid_ratio"], out["elimination"]], np.float64))
