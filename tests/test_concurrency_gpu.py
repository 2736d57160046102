"""Scene concurrency: eight host threads, each with its own HIP stream, push 200 tiny scenes through
the whole path (prefetch -> coordinate plan -> network -> head -> vote -> decode -> NMS).  The containers the threads
share (pinned landing buffers, prefetch entries, per-stream scratch, range flags, the C side's plan-side pool) are
locked, and EVERY stage's output - network output, head outputs, grid origin and shape, the vote grids, the candidate
cells, the detections - must equal the one-at-a-time result exactly.

This test found the round-3 vote bug: with the 16 x 32-cell / 8-wave tile workgroups of rounds 1-2, a third of the vote
launches that overlapped another stream's fp16 matrix-core convolutions put a few dozen cells of one tile wrong
(profiles/r3/vote_concurrency_findings.txt); the 32 x 32 / 16-wave shape is exact."""
import os
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eight_threads_two_hundred_scenes_each(cuda):
    from canonicalvoting_amd import pipeline
    from canonicalvoting_amd.hough import HoughVoting
    from canonicalvoting_amd.minkunet import MinkUNet34C
    from canonicalvoting_amd.synth import make_scene
    dev = cuda
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).to(dev).eval()
    scenes = []
    for seed in range(4):
        sc = make_scene(seed, n_points=1500 + 250 * seed, res=0.06, room=(2.0, 1.0, 2.0), n_boxes=3, margin=0.6, box_scale=0.5)
        c4 = torch.cat([torch.zeros((len(sc.coords), 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
        f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
        scenes.append((c4, f))

    from canonicalvoting_amd import _lib, decode, hv_cuda
    from canonicalvoting_amd import me as ME

    def run(hv, k):
        """pipeline.detect_scene stage by stage, with a fingerprint of every stage's output"""
        c4, f = scenes[k % len(scenes)]
        with torch.no_grad():
            pts = (c4[:, 1:] * 0.06).float().contiguous()
            hv_cuda.prefetch_geometry(pts)
            y = model(ME.SparseTensor(f, c4, device=dev))
            xyz, scale, prob, cls = pipeline.head_joint(y.F)
            g = hv(pts, xyz, scale, prob)
        raw = decode.decode_boxes(g[0], g[1], g[2], pts, xyz, prob, cls, 0.06, thresh_high=5)
        dets = decode.nms_per_class(raw["boxes"], raw["scores"], raw["classes"])
        return (len(raw["cand_idx"]), [(c, round(float(s), 6)) for c, _, s in dets], float(y.F.abs().sum()),
                tuple(g[0].shape), float(g[0].double().sum()), int((g[0] >= 5).sum()), float(g[2].double().abs().sum()),
                [int(v) for v in raw["cand_idx"]], float(xyz.double().sum()), float(scale.double().sum()),
                float(prob.double().sum()), tuple(hv_cuda.recent_corner(g[0])), int((g[0] != 0).sum()))

    ref = [run(HoughVoting(0.06, 120), k) for k in range(len(scenes))]
    T, N = int(os.environ.get("CV_STRESS_T", "8")), 200
    errors, bad = [], []

    def worker(i):
        try:
            torch.cuda.set_device(dev)
            hv = HoughVoting(0.06, 120)
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                for k in range(N):
                    got = run(hv, k + i)
                    want = ref[(k + i) % len(scenes)]
                    if got != want:
                        bad.append((i, k, got, want))
            stream.synchronize()
            _lib.release_scratch(dev, stream)      # a retired stream gives its scratch back
        except BaseException as e:           # noqa: BLE001 - reported below
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    held = _lib.scratch_bytes()
    _lib.release_scratch(dev, torch.cuda.current_stream(dev))
    assert _lib.scratch_bytes() < held or held == 0      # only the main stream's buffers were left, and they go too
    names = ["n_cand", "dets", "y_abs_sum", "grid_shape", "grid_obj_sum", "cells_ge_5", "grid_scale_abs_sum", "cand_idx",
             "xyz_sum", "scale_sum", "prob_sum", "corner", "touched_cells"]
    diff = {}
    for _, _, got, want in bad:
        for nm, a, b in zip(names, got, want):
            if a != b:
                diff.setdefault(nm, []).append((a, b))
    print("bad %d of %d; differing fields:" % (len(bad), T * N), {k: (len(v), v[:2]) for k, v in diff.items()})
    assert not bad, (len(bad), {k: (len(v), v[:2]) for k, v in diff.items()})
