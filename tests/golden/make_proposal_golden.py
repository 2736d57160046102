"""Golden vectors for the SUN RGB-D proposal sampler, produced by executing the reference's own
lines (sunrgbd/brnetcanon.py:86-91 unravel_index, :104-162 class HoughVotingModule) on CPU torch in the build
container.  The vote the module calls (HVFunction.apply -> hv_cuda.forward with the 7th `corners` argument) is bound to
the CPU oracle's vote; `device='cuda'` in the constructor is dropped; torch.multinomial is wrapped so the draws of
every loop trip are recorded (the only stochastic step, :137).  Stores the draws, the candidates, scales and probs
(tests/golden/proposal_ref.npz); inputs are regenerated from the seed by `make_inputs`.

    python tests/golden/make_proposal_golden.py            # needs /root/reference
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/sunrgbd/brnetcanon.py"
RES, ROTS, NPROP = 0.05, 60, 128


def make_inputs(seed=5):
    from canonicalvoting_amd.synth import make_scene, synth_predictions
    sc = make_scene(seed, n_points=6000, res=RES, room=(4.0, 2.4, 4.0), n_boxes=5)
    xyz, scale, prob, _ = synth_predictions(sc)
    pts = sc.points.astype(np.float32)
    corners = np.stack([pts.min(0), pts.max(0)]).astype(np.float32)
    rng = np.random.default_rng(seed)
    votes = (pts[rng.choice(len(pts), 256, replace=False)] + rng.normal(0, 0.05, (256, 3))).astype(np.float32)
    return pts, xyz, scale, prob, corners, votes


def ref_lines(a, b):
    src = open(REF).read().splitlines()
    return textwrap.dedent("\n".join(src[a - 1:b])) + "\n"


if __name__ == "__main__":
    assert os.path.exists(REF), "run where /root/reference is mounted"
    import oracle
    pts, xyz, scale, prob, corners, votes = make_inputs()
    draws = []

    class TorchOnCpu:
        """the torch module with `device=` dropped from tensor() and multinomial() recorded"""
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def tensor(*a, device=None, **k):
            return torch.tensor(*a, **k)

        @staticmethod
        def multinomial(*a, **k):
            s = torch.multinomial(*a, **k)
            draws.append(s.numpy().copy())
            return s

    class HVFunction:
        @staticmethod
        def apply(points, xyz_, scale_, obj, res, num_rots, corners_):
            g = oracle.hv_forward(points.numpy(), xyz_.numpy(), scale_.numpy(), obj.numpy(), float(res), int(num_rots),
                                  corners=corners_.numpy())
            return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in g)

    ns = {"torch": TorchOnCpu(), "nn": torch.nn, "HVFunction": HVFunction}
    exec(ref_lines(86, 91), ns)
    exec(ref_lines(104, 162), ns)
    hv = ns["HoughVotingModule"](res=RES, nms_size=0.3, thresh=0, num_proposal=NPROP, num_rots=ROTS)
    t = torch.from_numpy
    torch.manual_seed(0)
    cand, probs, scales = hv(t(pts), t(xyz), t(scale), t(prob), t(corners), t(votes))
    np.savez_compressed(os.path.join(HERE, "proposal_ref.npz"), seed=5, draws=np.stack(draws), candidates=cand.numpy(),
                        probs=probs.numpy(), scales=scales.numpy())
    print("trips", len(draws), cand.shape, scales.shape, float(probs.abs().max()))
