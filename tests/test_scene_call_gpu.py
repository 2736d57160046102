"""cv_detect_scene_f32 - one C call per scene (coordinate plan -> network program -> head -> vote -> decode -> NMS) -
against the call-by-call pipeline: the same kernels in the same order, so EVERY output is the same bits (network
output, head outputs, the three vote grids, candidate cells, verdicts, boxes, scores, classes, detections), on the
network's own predictions and on teacher predictions, one scene at a time and from several host threads."""
import threading

import numpy as np
import pytest
import torch

from canonicalvoting_amd import decode, pipeline
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene, synth_predictions

pytestmark = pytest.mark.gpu


def resident(seed, n, cuda, small):
    kw = dict(res=0.06, room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5) if small else {}
    sc = make_scene(seed, n_points=n, **kw)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    c4 = torch.cat([torch.zeros((n, 1), dtype=torch.int32), torch.from_numpy(sc.coords).int()], 1).to(cuda)
    feats = (t(sc.feats) * 2 - 1).contiguous()
    pts = (c4[:, 1:] * sc.res).float().contiguous()
    xyz, scale, prob, cls = [t(a) for a in synth_predictions(sc)]
    return sc, c4, feats, pts, (xyz, scale, prob, cls.int())


def by_calls(model, hv, c4, feats, pts, res, teacher, thresh_high, policy=None):
    with torch.no_grad(), pipeline.scene_policy(policy):
        x = ME.SparseTensor(feats, c4, device=feats.device)
        y = model(x).F
        pred = pipeline.head_joint(y)
        use = teacher if teacher is not None else pred
        g = hv(pts, use[0], use[1], use[2])
    raw = decode.decode_boxes(g[0], g[1], g[2], pts, use[0], use[2], use[3], res, thresh_high=thresh_high)
    dets = decode.nms_per_class(raw["boxes"], raw["scores"], raw["classes"])
    return dict(y=y, pred=pred, grids=g, raw=raw, dets=dets)


def assert_same(a, keep, dets, raw, y, tag):
    assert torch.equal(a["y"], y), tag + ": network output"
    for u, v, name in zip(a["pred"], keep["net_pred"], ("xyz", "scale", "prob", "class")):
        assert torch.equal(u, v), tag + ": head " + name
    for u, v, name in zip(a["grids"], keep["grids"], ("obj", "rot", "scale")):
        assert u.shape == v.shape and torch.equal(u, v), tag + ": grid_" + name
    for k in ("cand_idx", "verdict", "boxes", "scores", "classes"):
        assert np.array_equal(a["raw"][k], raw[k]), tag + ": " + k
    assert len(a["dets"]) == len(dets)
    for (c0, b0, s0), (c1, b1, s1) in zip(a["dets"], dets):
        assert c0 == c1 and s0 == s1 and np.array_equal(b0, b1), tag + ": detections"


@pytest.mark.parametrize("in_flight", [None, 7])
@pytest.mark.parametrize("n,small,thresh", [(3000, True, 20), (80000, False, 60)])
def test_one_call_scene_equals_the_call_by_call_pipeline(cuda, built_lib, n, small, thresh, in_flight):
    """in_flight None: no policy (the thread's / process-wide launch sizing); 7: bench.py's timed region - the policy travels in
    cv_scene_desc for the one call and in the thread's values for the call-by-call pipeline, and both give the same bits"""
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).to(cuda).eval()
    sc, c4, feats, pts, teacher = resident(1, n, cuda, small)
    hv = HoughVoting(sc.res, 120)
    policy = None if in_flight is None else pipeline.policy_for_scenes_in_flight(in_flight)
    for pred, tag in ((teacher, "teacher predictions"), (None, "network predictions")):
        want = by_calls(model, hv, c4, feats, pts, sc.res, pred, thresh, policy)
        for rep in range(2):            # the second call runs on the grown scratch
            keep = {}
            dets, raw, y = pipeline.detect_scene_c(model, hv, c4, feats, sc.res, scan_points=pts, predictions=pred,
                                                   keep=keep, thresh_high=thresh, policy=policy)
            assert_same(want, keep, dets, raw, y, "%s (%d points, call %d)" % (tag, n, rep))
        if pred is not None:
            assert len(want["raw"]["boxes"]) >= 2          # the comparison saw accepted boxes and rejected candidates
            assert len(want["raw"]["cand_idx"]) > len(want["raw"]["boxes"])
    # the stage events of the call are recorded in order on the scene's stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    pipeline.detect_scene_c(model, hv, c4, feats, sc.res, scan_points=pts, predictions=teacher, events=ev, thresh_high=thresh)
    torch.cuda.synchronize()
    times = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    assert all(t > 0 for t in times) and times[0] > times[1], times          # the network is the longest stage, the head the shortest


def test_two_hosts_with_different_policies_in_one_process(cuda, built_lib):
    """the launch sizing is per call.  Two threads run the same 80k scene at the same time, one under the
    one-scene policy and one under the seven-in-flight policy; each must get exactly what it gets when it runs alone (the
    network output differs between the policies in fp32 summation order - a thread that picked up the other's sizing shows)."""
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).to(cuda).eval()
    sc, c4, feats, pts, teacher = resident(1, 80000, cuda, False)
    pols = [pipeline.policy_for_scenes_in_flight(1), pipeline.policy_for_scenes_in_flight(7)]
    alone = []
    for pol in pols:
        keep = {}
        pipeline.detect_scene_c(model, HoughVoting(sc.res, 120), c4, feats, sc.res, scan_points=pts, predictions=teacher, keep=keep, policy=pol)
        alone.append(keep)
    assert not torch.equal(alone[0]["y"], alone[1]["y"])                       # the policies ARE visible in the summation order
    assert all(torch.equal(a, b) for a, b in zip(alone[0]["grids"], alone[1]["grids"]))     # ... and only there
    errors, start = [], threading.Barrier(2)

    def worker(i):
        try:
            torch.cuda.set_device(cuda)
            hv = HoughVoting(sc.res, 120)
            with torch.cuda.stream(torch.cuda.Stream(cuda)):
                start.wait()
                for k in range(6):
                    keep = {}
                    if k % 2:                       # the call-by-call path under the thread's values
                        with torch.no_grad(), pipeline.scene_policy(pols[i]):
                            y = model(ME.SparseTensor(feats, c4, device=cuda)).F
                    else:
                        pipeline.detect_scene_c(model, hv, c4, feats, sc.res, scan_points=pts, predictions=teacher, keep=keep, policy=pols[i])
                        y = keep["y"]
                    assert torch.equal(y, alone[i]["y"]), "thread %d step %d ran under another launch policy" % (i, k)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:      # noqa: BLE001
            errors.append(repr(e))
            start.abort()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]
    from canonicalvoting_amd import _lib
    L = _lib.lib()
    assert L.cv_sp_set_split_target_thread(0) == 0 and L.cv_hv_set_part_records_thread(0) == 0 and ME.masked_min_rows() == ME.CoordinateManager.MASKED_MIN_ROWS


def test_one_call_scene_from_four_threads(cuda, built_lib):
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).to(cuda).eval()
    scenes = [resident(10 + k, 2500 + 300 * k, cuda, True) for k in range(3)]
    hv0 = HoughVoting(0.06, 120)
    want = [by_calls(model, hv0, s[1], s[2], s[3], 0.06, s[4], 20) for s in scenes]
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(cuda)
            hv = HoughVoting(0.06, 120)
            with torch.cuda.stream(torch.cuda.Stream(cuda)):
                for k in range(30):
                    j = (k + i) % len(scenes)
                    s = scenes[j]
                    keep = {}
                    dets, raw, y = pipeline.detect_scene_c(model, hv, s[1], s[2], 0.06, scan_points=s[3], predictions=s[4],
                                                           keep=keep, thresh_high=20)
                    assert_same(want[j], keep, dets, raw, y, "thread %d step %d" % (i, k))
                torch.cuda.current_stream().synchronize()
        except BaseException as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]
