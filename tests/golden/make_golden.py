"""Generates tests/golden/*.npz from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py

The reference ships no golden vectors and none of its hot path can run here
(SURVEY.md 8c), so these fixtures are minted by oracle/ (C restatement, cross-checked
against oracle/hv_numpy.py and the analytic known-answer tests in tests/).  They pin
the oracle against regressions and travel to the GPU box for the HIP parity tests.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from canonicalvoting_amd.synth import make_scene, synth_predictions  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def vote_case(name, seed, n, room, n_boxes, num_rots=120, res=0.03, thresh=60.0, full=True, **kw):
    sc = make_scene(seed, n_points=n, res=res, room=room, n_boxes=n_boxes, **kw)
    xyz, scale, prob, cls = synth_predictions(sc)
    pts = sc.points
    g_obj, g_rot, g_scale, vin = oracle.hv_forward(pts, xyz, scale, prob, res, num_rots, return_vin=True)
    rng = np.random.default_rng(seed + 77)
    grad = rng.normal(0, 1, g_obj.shape).astype(np.float32)
    d_xyz, d_scale, d_obj = oracle.hv_backward(grad, pts, xyz, scale, prob, res, num_rots)
    corner, _, dims = oracle.grid_geometry(pts, res)
    dec = oracle.decode(g_obj, g_rot, g_scale, corner, res, pts, xyz, prob, cls,
                        oracle.DecodeParams.default(thresh_high=thresh))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), coords=sc.coords, res=np.float32(res),
        num_rots=np.int32(num_rots), xyz=xyz, scale=scale, prob=prob, cls=cls, dims=np.array(dims),
        corner=corner, v_in=np.int64(vin), grid_obj=g_obj,
        grid_rot=g_rot if full else np.zeros(0, np.float32),
        grid_scale=g_scale if full else np.zeros(0, np.float32),
        rot_sum=g_rot.astype(np.float64).sum((0, 1, 2)), scale_sum=g_scale.astype(np.float64).sum((0, 1, 2)),
        grad=grad, d_xyz=d_xyz, d_scale=d_scale, d_obj=d_obj,
        dec_thresh_high=np.float32(thresh), dec_cand=dec["cand_idx"], dec_verdict=dec["verdict"],
        dec_boxes=dec["boxes"], dec_scores=dec["scores"], dec_classes=dec["classes"])
    print(name, "N", n, "dims", dims, "v_in", vin, "cands", len(dec["cand_idx"]), "boxes",
          len(dec["boxes"]), "max", float(g_obj.max()))


if __name__ == "__main__":
    # small rooms keep the fixtures at a few hundred KB
    vote_case("vote_512", seed=3, n=512, room=(1.5, 0.9, 1.5), n_boxes=2, num_rots=24, res=0.06,
              margin=0.5, box_scale=0.4, thresh=8.0)
    vote_case("vote_2k", seed=5, n=2048, room=(2.0, 1.0, 2.0), n_boxes=3, num_rots=120, res=0.05,
              margin=0.6, box_scale=0.5, thresh=60.0)
