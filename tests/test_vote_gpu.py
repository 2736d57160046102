"""HIP vote op vs the CPU oracle, through the hv_cuda drop-in (which calls the C ABI).

Bar (BASELINE.json north_star): grid shape, in-bounds vote count and the set of touched cells
are exact; accumulated floats agree within 1e-4 relative (fp32 atomic summation order is the
only difference, as it is run-to-run in the reference itself)."""
import os

import numpy as np
import pytest
import torch

import oracle
from canonicalvoting_amd import hv_cuda
from canonicalvoting_amd.hough import HoughVoting
from canonicalvoting_amd.synth import make_scene, synth_predictions

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-4      # north_star float tolerance


def dev_inputs(cuda, pts, xyz, scale, prob):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    return t(pts), t(xyz), t(scale), t(prob)


def run_hip(cuda, pts, xyz, scale, prob, res, R, algo):
    hv_cuda.set_algorithm(algo)
    try:
        hv = HoughVoting(res, R)
        with torch.no_grad():
            out = hv(*dev_inputs(cuda, pts, xyz, scale, prob))
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in out]
    finally:
        hv_cuda.set_algorithm(0)


QUANTUM = 2.0 ** -36     # fixed-point quantum of the tile kernel's quotient numerators (hv_vote.hip)


def assert_grids_close(hip, ref, tag="", inputs=None, counts=None):
    """Every cell of the three grids, no live-cell mask.  grid_obj: 1e-4 relative.  Quotient grids
    (x / (w + 1e-7), hv_cuda_kernel.cu:112-117) on EVERY touched cell: |hip - ref| <= 1e-4 max(1, |ref|) +
    n_c * 2^-36 / (w_c + 1e-7), n_c the number of contributions the cell received - each contribution to a numerator is
    rounded once to the 2^-36 quantum (<= 2^-37 each) where the reference's fp32 atomics round relative to the running sum,
    so the bound on the quotient scales with 1 / weight and vanishes on every cell of non-negligible weight.
    inputs = (points, xyz, scale, res, num_rots[, corner]) supplies n_c (oracle/hv_numpy.contribution_counts)."""
    g_obj, g_rot, g_scale = hip
    r_obj, r_rot, r_scale = ref
    assert g_obj.shape == r_obj.shape and g_rot.shape == r_rot.shape and g_scale.shape == r_scale.shape, tag
    assert np.array_equal(g_obj == 0, r_obj == 0), tag + ": set of touched cells differs"
    scale_ = max(1.0, float(np.abs(r_obj).max()))
    np.testing.assert_allclose(g_obj, r_obj, rtol=RTOL, atol=1e-6 * scale_, err_msg=tag)
    if counts is None:
        from oracle import hv_numpy
        pts, xyz, scale, res, R = inputs[:5]
        corner = inputs[5] if len(inputs) > 5 else None
        counts = hv_numpy.contribution_counts(pts, xyz, scale, res, R, corner, list(r_obj.shape) if corner is not None else None)
    assert counts.shape == r_obj.shape
    assert (counts[r_obj != 0] > 0).all(), tag                              # a touched cell has contributions
    slack = (counts.astype(np.float64) * QUANTUM / (r_obj.astype(np.float64) + 1e-7))[..., None]
    for g, r, name in ((g_rot, r_rot, "grid_rot"), (g_scale, r_scale, "grid_scale")):
        tol = RTOL * np.maximum(1.0, np.abs(r.astype(np.float64))) + slack
        bad = np.abs(g.astype(np.float64) - r.astype(np.float64)) > tol
        assert not bad.any(), "%s %s: %d cells beyond the bound, worst %g at weight %g" % (
            tag, name, int(bad.sum()), float(np.abs(g - r)[bad].max()), float(np.broadcast_to(r_obj[..., None], g.shape)[bad].min()))
    dead = r_obj == 0
    assert not g_rot[dead].any() and not g_scale[dead].any(), tag


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("seed,n,R", [(0, 600, 24), (1, 2048, 120), (2, 777, 60), (3, 64, 7)])
def test_forward_matches_oracle_small(cuda, built_lib, algo, seed, n, R):
    sc = make_scene(seed, n_points=n, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5,
                    box_scale=0.4)
    xyz, scale, prob, _ = synth_predictions(sc)
    ref = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, R, return_vin=True)
    hip = run_hip(cuda, sc.points, xyz, scale, prob, sc.res, R, algo)
    assert_grids_close(hip, ref[:3], "algo %d seed %d" % (algo, seed), inputs=(sc.points, xyz, scale, sc.res, R))
    corner, _, dims = oracle.grid_geometry(sc.points, sc.res)
    p, x, s, _ = dev_inputs(cuda, sc.points, xyz, scale, prob)
    assert hv_cuda.count_votes(p, x, s, sc.res, R, corner, dims) == ref[3]


@pytest.mark.parametrize("algo", [1, 2])
def test_forward_matches_oracle_8k_scannet_res(cuda, built_lib, algo):
    """BASELINE config 1 size (8k points, res 0.03, 120 rotations, shifted origin)."""
    sc = make_scene(11, n_points=8000)
    xyz, scale, prob, _ = synth_predictions(sc)
    ref = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, 120)
    hip = run_hip(cuda, sc.points, xyz, scale, prob, sc.res, 120, algo)
    assert_grids_close(hip, ref, "8k algo %d" % algo, inputs=(sc.points, xyz, scale, sc.res, 120))


@pytest.mark.parametrize("name", ["vote_512", "vote_2k"])
def test_forward_backward_match_golden(cuda, built_lib, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    res, R = float(z["res"]), int(z["num_rots"])
    pts = (z["coords"].astype(np.float32) * np.float32(res)).astype(np.float32)
    for algo in (1, 2):
        g_obj, g_rot, g_scale = run_hip(cuda, pts, z["xyz"], z["scale"], z["prob"], res, R, algo)
        assert list(g_obj.shape) == list(z["dims"])
        # cell by cell against the stored grids (all three, every touched cell)
        assert_grids_close((g_obj, g_rot, g_scale), (z["grid_obj"], z["grid_rot"], z["grid_scale"]),
                           "%s algo %d" % (name, algo), inputs=(pts, z["xyz"], z["scale"], res, R))
    p, x, s, o = dev_inputs(cuda, pts, z["xyz"], z["scale"], z["prob"])
    hv = HoughVoting(res, R)
    d = hv_cuda.backward(torch.from_numpy(z["grad"]).to(cuda), p, x, s, o, hv.res, hv.num_rots)
    for got, key in zip(d, ("d_xyz", "d_scale", "d_obj")):
        ref = z[key]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=RTOL * max(1.0, np.abs(ref).max()))


def test_backward_through_autograd(cuda, built_lib):
    """HVFunction.backward (eval_joint.py:32-38): only grad_obj is propagated."""
    sc = make_scene(5, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
    xyz, scale, prob, _ = synth_predictions(sc)
    p, x, s, o = dev_inputs(cuda, sc.points, xyz, scale, prob)
    x.requires_grad_(True); s.requires_grad_(True); o.requires_grad_(True)
    hv = HoughVoting(sc.res, 36)
    g_obj, g_rot, g_scale = hv(p, x, s, o)
    rng = np.random.default_rng(0)
    grad = rng.normal(0, 1, tuple(g_obj.shape)).astype(np.float32)
    (g_obj * torch.from_numpy(grad).to(cuda)).sum().backward()
    r_xyz, r_scale, r_obj = oracle.hv_backward(grad, sc.points, xyz, scale, prob, sc.res, 36)
    for got, ref in ((x.grad, r_xyz), (s.grad, r_scale), (o.grad, r_obj)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=RTOL, atol=RTOL * max(1.0, np.abs(ref).max()))
    # grad_rot / grad_scale are ignored by the reference's backward
    x.grad = None
    g_obj, g_rot, g_scale = hv(p, x, s, o)
    (g_rot.sum() + g_scale.sum()).backward()
    assert float(x.grad.abs().max()) == 0.0


def test_edge_cases(cuda, built_lib):
    hv = HoughVoting(0.03, 120)
    # degenerate cloud: one cell grid, every vote fails `>= size-1` (hv_cuda_kernel.cu:41-44)
    p = torch.zeros(5, 3, device=cuda); x = torch.zeros(5, 3, device=cuda)
    s = torch.ones(5, 3, device=cuda); o = torch.ones(5, device=cuda)
    for algo in (1, 2):
        hv_cuda.set_algorithm(algo)
        g = hv(p, x, s, o)
        assert tuple(g[0].shape) == (1, 1, 1) and float(g[0].sum()) == 0.0
    hv_cuda.set_algorithm(0)
    # two points: xyz = 0 votes land on the nodes; the max corner is out of bounds
    p = torch.tensor([[0, 0, 0], [0.3, 0.3, 0.3]], device=cuda)
    g = hv(p, torch.zeros(2, 3, device=cuda), torch.ones(2, 3, device=cuda), torch.ones(2, device=cuda))
    ref = oracle.hv_forward(p.cpu().numpy(), np.zeros((2, 3)), np.ones((2, 3)), np.ones(2), 0.03, 120)
    assert g[0].shape == ref[0].shape
    np.testing.assert_allclose(g[0].cpu().numpy(), ref[0], rtol=1e-6)
    assert float(g[0][0, 0, 0]) == 120.0
    # outputs are fresh, writable and independent (callers zero grid_obj in place, eval_joint.py:211)
    g[0][0:1, 0:1, 0:1] = 0
    assert float(g[0][0, 0, 0]) == 0.0


def test_input_checks_match_reference_messages(cuda, built_lib):
    hv = HoughVoting(0.03, 120)
    p = torch.rand(10, 3, device=cuda); x = torch.rand(10, 3, device=cuda)
    s = torch.rand(10, 3, device=cuda); o = torch.rand(10, device=cuda)
    with pytest.raises(RuntimeError, match="points must be a CUDA tensor"):
        hv_cuda.forward(p.cpu(), x, s, o, hv.res, hv.num_rots)
    with pytest.raises(RuntimeError, match="xyz_labels must be contiguous"):
        hv_cuda.forward(p, torch.rand(3, 10, device=cuda).t(), s, o, hv.res, hv.num_rots)
    with pytest.raises(RuntimeError, match="res must be a CUDA tensor"):
        hv_cuda.forward(p, x, s, o, hv.res.cpu(), hv.num_rots)
    with pytest.raises(RuntimeError, match="grad_grid must be a CUDA tensor"):
        hv_cuda.backward(torch.zeros(2, 2, 2), p, x, s, o, hv.res, hv.num_rots)
    with pytest.raises(RuntimeError):
        hv_cuda.forward(p[:0], x[:0], s[:0], o[:0], hv.res, hv.num_rots)      # N = 0


def test_corners_variant(cuda, built_lib):
    """7-argument forward (sunrgbd/brnetcanon.py:99): grid origin/extent from corners[2,3]."""
    sc = make_scene(9, n_points=1000, res=0.05, room=(2.0, 1.0, 2.0), n_boxes=2, margin=0.6, box_scale=0.5)
    xyz, scale, prob, _ = synth_predictions(sc)
    pts = sc.points
    corners = np.stack([pts.min(0) - 0.2, pts.max(0) + 0.3]).astype(np.float32)
    ref = oracle.hv_forward(pts, xyz, scale, prob, 0.05, 60, corners=corners)
    hv = HoughVoting(0.05, 60)
    out = hv(*dev_inputs(cuda, pts, xyz, scale, prob), corners=torch.from_numpy(corners).to(cuda))
    assert_grids_close([o.cpu().numpy() for o in out], ref, "corners", inputs=(pts, xyz, scale, 0.05, 60, corners[0]))


@pytest.mark.parametrize("algo", [1, 2])
def test_full_size_80k_properties(cuda, built_lib, algo):
    """BASELINE config 2 size.  Size-independent properties (the oracle takes ~1 s here so it is
    also compared directly): mass conservation (sum grid_obj = sum over in-bounds votes of obj,
    because trilinear weights sum to 1) and linearity in obj."""
    sc = make_scene(0, n_points=80000)
    xyz, scale, prob, _ = synth_predictions(sc)
    hip = run_hip(cuda, sc.points, xyz, scale, prob, sc.res, 120, algo)
    ref = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, 120)
    assert_grids_close(hip, ref, "80k algo %d" % algo, inputs=(sc.points, xyz, scale, sc.res, 120))
    hip2 = run_hip(cuda, sc.points, xyz, scale, (prob * 2).astype(np.float32), sc.res, 120, algo)
    np.testing.assert_allclose(hip2[0], 2 * hip[0], rtol=RTOL, atol=1e-4)
    np.testing.assert_allclose(hip2[0].sum(dtype=np.float64), 2 * ref[0].sum(dtype=np.float64), rtol=1e-5)


def test_huge_scale_contributions_take_the_exact_slow_path(cuda, built_lib):
    """the fixed-point accumulation converts contributions with a magic-number trick valid below 2^14; a diverged
    scale head (exp of a large logit) must still give the oracle's sums"""
    sc = make_scene(13, n_points=1500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
    xyz, scale, prob, _ = synth_predictions(sc)
    scale = scale.copy()
    scale[::7] *= 1.0e5                      # w * s up to ~1e5: above the fast-path bound
    xyz = (xyz * 1e-5 * (scale > 1e3) + xyz * (scale <= 1e3)).astype(np.float32)   # keep those votes in bounds
    ref = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, 24)
    hip = run_hip(cuda, sc.points, xyz, scale, prob, sc.res, 24, 2)
    assert np.array_equal(hip[0] == 0, ref[0] == 0)
    np.testing.assert_allclose(hip[0], ref[0], rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(ref[0]).max())))
    live = ref[0] > 1e-3 * max(1.0, float(np.abs(ref[0]).max()))
    np.testing.assert_allclose(hip[2][live], ref[2][live], rtol=1e-4, atol=1e-4)
    assert float(np.abs(ref[2]).max()) > 2e4                    # the slow path was exercised


def test_kernel_events_bracket_the_accumulation_kernel(cuda, built_lib):
    """cv_hv_set_kernel_events (bench.py's `roofline` timing): the two events are recorded around hv_fwd_tiles on the
    stream of the call - a positive time below the whole op's, the grids unchanged, nothing recorded once switched off
    or on another thread."""
    import threading
    from canonicalvoting_amd import _lib
    L = _lib.lib()
    sc = make_scene(3, n_points=20000)
    xyz, scale, prob, _ = synth_predictions(sc)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    args = (t(sc.points), t(xyz), t(scale), t(prob))
    hv = HoughVoting(sc.res, 120)
    plain = [g.clone() for g in hv(*args)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for e in ev:
        e.record()                                       # creates the hipEvent_t handles
    assert all(e.cuda_event for e in ev)
    torch.cuda.synchronize()
    assert L.cv_hv_set_kernel_events(ev[1].cuda_event, ev[2].cuda_event) == 0
    try:
        ev[0].record()
        timed = hv(*args)
        ev[3].record()
        torch.cuda.synchronize()
        k_ms, op_ms = ev[1].elapsed_time(ev[2]), ev[0].elapsed_time(ev[3])
        assert 0.02 < k_ms < op_ms, (k_ms, op_ms)
        assert ev[0].elapsed_time(ev[1]) >= 0 and ev[2].elapsed_time(ev[3]) >= 0
        for a, b in zip(plain, timed):
            assert torch.equal(a, b)
        # thread-local: another thread's call leaves the events alone
        before = ev[1].elapsed_time(ev[2])
        th = threading.Thread(target=lambda: (hv(*args), torch.cuda.synchronize()))
        th.start(); th.join()
        assert ev[1].elapsed_time(ev[2]) == before
    finally:
        L.cv_hv_set_kernel_events(None, None)
    hv(*args)
    torch.cuda.synchronize()
    assert ev[1].elapsed_time(ev[2]) == before


@pytest.mark.parametrize("in_flight", [1, 7])
@pytest.mark.parametrize("points,large", [(80000, False), (300000, True)])
def test_production_size_grids_are_the_same_bits_on_every_run(cuda, built_lib, points, large, in_flight):
    """hot (plane, tile) pairs are split into parts whose membership follows the atomic order of the
    scatter / list passes; the parts are published as raw 2^-36 fixed-point words and added as integers, so all three
    grids must be bit-identical run to run at the sizes where parts exist (80k: streaming launch, up to 8 parts per
    plane; 300k: work queue, parts by weight), also while other streams keep the chip busy."""
    kw = dict(room=(9.0, 3.0, 9.0), n_boxes=40) if large else {}
    sc = make_scene(2, n_points=points, **kw)
    xyz, scale, prob, _ = synth_predictions(sc)
    args = dev_inputs(cuda, sc.points, xyz, scale, prob)
    hv = HoughVoting(sc.res, 120)
    from canonicalvoting_amd import pipeline
    # (both launch sizings of bench.py: the library's, and the timed region's 12288 records per part)
    with torch.no_grad(), pipeline.scene_policy(pipeline.policy_for_scenes_in_flight(in_flight)):
        first = [g.clone() for g in hv(*args)]
        side = torch.cuda.Stream()
        a = torch.randn(2048, 2048, device=cuda)
        for rep in range(8):
            with torch.cuda.stream(side):              # uneven load next to the vote: changes who arrives last
                for _ in range(rep % 3):
                    a = (a @ a).clamp_(-1, 1)
            again = hv(*args)
            for name, x, y in zip(("obj", "rot", "scale"), first, again):
                assert torch.equal(x, y), "grid_%s differs on repetition %d (%d cells)" % (name, rep, int((x != y).sum()))
    torch.cuda.synchronize()
    assert float(first[0].max()) > 60.0                # peaked maps: the hot planes that get split exist


def test_compiled_extension_equals_the_ctypes_module_and_the_goldens(cuda, built_lib):
    """the compiled `hv_cuda` (csrc/hv_cuda_ext.cpp) and the ctypes `hv_cuda` call the same C ABI: forward and backward
    bit-identical on a seeded scene, incl. the 7-argument corners variant, on a non-default stream; the reference's error
    strings; and eval_joint.py's HVFunction / HoughVoting (lines 24-57) run unchanged over it."""
    from canonicalvoting_amd import hv_cuda_ext
    ext = hv_cuda_ext.load()
    sc = make_scene(5, n_points=6000)
    xyz, scale, prob, _ = synth_predictions(sc)
    args = dev_inputs(cuda, sc.points, xyz, scale, prob)
    hv = HoughVoting(sc.res, 120)
    with torch.no_grad():
        want = hv_cuda.forward(*args, hv.res, hv.num_rots)
        got = ext.forward(*args, hv.res, hv.num_rots)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got_side = ext.forward(*args, hv.res, hv.num_rots)
        side.synchronize()
    for a, b, c in zip(want, got, got_side):
        assert a.shape == b.shape and torch.equal(a, b) and torch.equal(a, c)
    assert got[0].data_ptr() != want[0].data_ptr() and got[0].is_contiguous()
    ref = oracle.hv_forward(sc.points, xyz, scale, prob, sc.res, 120)
    assert_grids_close([g.cpu().numpy() for g in got], ref, "compiled ext", inputs=(sc.points, xyz, scale, sc.res, 120))
    gg = torch.rand_like(want[0])
    for a, b in zip(hv_cuda.backward(gg, *args, hv.res, hv.num_rots), ext.backward(gg, *args, hv.res, hv.num_rots)):
        assert torch.equal(a, b)
    corners = torch.from_numpy(np.stack([sc.points.min(0) - 0.2, sc.points.max(0) + 0.3]).astype(np.float32)).to(cuda)
    for a, b in zip(hv_cuda.forward(*args, hv.res, hv.num_rots, corners), ext.forward(*args, hv.res, hv.num_rots, corners)):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError, match="xyz_labels must be contiguous"):
        ext.forward(args[0], torch.rand(3, len(sc.points), device=cuda).t(), args[2], args[3], hv.res, hv.num_rots)
    with pytest.raises(RuntimeError, match="res must be a CUDA tensor"):
        ext.forward(*args, hv.res.cpu(), hv.num_rots)
    with pytest.raises(RuntimeError):
        ext.forward(args[0][:0], args[1][:0], args[2][:0], args[3][:0], hv.res, hv.num_rots)

    # the reference's autograd wrapper, as written at eval_joint.py:24-38, over the compiled module
    class HVFunction(torch.autograd.Function):
        @staticmethod
        def forward(ctx, points, xyz_labels, scale_labels, obj_labels, res, num_rots):
            outputs = ext.forward(points, xyz_labels, scale_labels, obj_labels, res, num_rots)
            ctx.save_for_backward(points, xyz_labels, scale_labels, obj_labels, res, num_rots)
            return tuple(outputs)

        @staticmethod
        def backward(ctx, grad_obj, grad_rot, grad_scale):
            d = ext.backward(grad_obj.contiguous(), *ctx.saved_tensors)
            return None, d[0], d[1], d[2], None, None

    x = args[1].clone().requires_grad_(True)
    HVFunction.apply(args[0], x, args[2], args[3], hv.res, hv.num_rots)[0].sum().backward()
    x2 = args[1].clone().requires_grad_(True)
    hv(args[0], x2, args[2], args[3])[0].sum().backward()
    assert torch.equal(x.grad, x2.grad)
