"""Neighbour windows (csrc/sparse_win.hip): the window plan is integer work - checked exactly against the kernel map it
was built from - and conv_win, the convolution that multiplies all 27 offsets out of a tile's window in LDS, against the
mask-sorted conv_hl path (itself checked against the oracle) and against a float64 gather-matmul of the same operands,
within the north_star tolerance.  Replaces the 3x3x3 MinkowskiConvolution calls of the fine levels
(utils/minkunet.py:122-180, utils/resnet.py:118-154)."""
import numpy as np
import pytest
import torch

from canonicalvoting_amd import _lib
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene

pytestmark = pytest.mark.gpu

WIN_T, WIN_CAP = 256, 512
NONE, OUT = WIN_CAP * 64, WIN_CAP * 64 + 64       # no neighbour / neighbour beyond the window (two rows of zeros)


def sorted_manager(cuda, coords4):
    """the spatially sorted twin (the row order the fused network - and the windows - work in)"""
    cm = ME.CoordinateManager(torch.from_numpy(coords4).to(cuda, torch.int32))
    return cm.fused_plan()[0]


def scene(seed, n, dense=False, batch=1):
    if batch > 1:   # several scans in one tensor (batch index in column 0, as the training collate does)
        return np.concatenate([np.concatenate([np.full((n, 1), b, np.int64), scene(seed + b, n)[:, 1:]], 1) for b in range(batch)])
    if n >= 200000:  # BASELINE config 5 shaped: a 9 x 3 x 9 m room
        sc = make_scene(seed, n_points=n, room=(9.0, 3.0, 9.0), n_boxes=40)
        return np.concatenate([np.zeros((n, 1), np.int64), sc.coords], 1)
    if dense:       # a solid block: every voxel has all 27 neighbours, windows overflow their capacity
        side = int(round(n ** (1 / 3)))
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
        return np.concatenate([np.zeros((len(g), 1), np.int64), g], 1)
    small = n < 10000
    sc = make_scene(seed, n_points=n, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4) if small \
        else make_scene(seed, n_points=n)
    return np.concatenate([np.zeros((n, 1), np.int64), sc.coords], 1)


def split_windows(win, n):
    ntiles = (n + WIN_T - 1) // WIN_T
    w = win.cpu().numpy()
    rows = w[:ntiles * WIN_CAP].reshape(ntiles, WIN_CAP)
    lm = w[ntiles * WIN_CAP:ntiles * (WIN_CAP + 27 * WIN_T // 2)].view(np.uint16).reshape(ntiles, 27, WIN_T)
    return rows, lm.transpose(0, 2, 1)                                   # [tile][row][offset]


@pytest.mark.parametrize("seed,n,ts,dense,batch", [(0, 700, 1, False, 1), (1, 40000, 1, False, 1), (2, 40000, 2, False, 1),
                                                    (3, 8000, 1, True, 1), (4, 80000, 1, False, 1), (5, 20000, 1, False, 3),
                                                    (6, 300000, 1, False, 1), (6, 300000, 4, False, 1)])
def test_window_plan_resolves_every_map_entry(cuda, built_lib, seed, n, ts, dense, batch):
    cm = sorted_manager(cuda, scene(seed, n, dense, batch))
    nbr = cm.kernel_map(3, ts).cpu().numpy()
    N = nbr.shape[0]
    rows, lm = split_windows(cm.windows(ts), N)
    assert _lib.lib().cv_sp_windows_words(N) == cm.windows(ts).numel()
    outside = pairs = 0
    for t in range(rows.shape[0]):
        m = nbr[t * WIN_T:(t + 1) * WIN_T]
        want = np.unique(m[m >= 0])                                     # ascending distinct input rows of the tile
        w = rows[t]
        k = min(len(want), WIN_CAP)
        assert np.array_equal(w[:k], want[:k]) and (w[k:] == -1).all()
        e = lm[t, :len(m)].astype(np.int64)                              # entry = slot * 64 + ((slot >> 2) & 3) * 16
        assert (lm[t, len(m):] == NONE).all()                            # rows beyond the level in the last tile
        assert np.array_equal(e == NONE, m < 0)
        inside = (m >= 0) & (e < NONE)
        slot = e[inside] >> 6
        assert np.array_equal(e[inside] & 63, ((slot >> 2) & 3) << 4)    # the row's read swizzle rides in bits 4-5
        assert np.array_equal(w[slot], m[inside])                        # the slot holds exactly the neighbour's row
        out = (m >= 0) & (e == OUT)
        assert np.isin(m[out], want[k:]).all()                          # only rows beyond the capacity are left outside
        outside += int(out.sum())
        pairs += int((m >= 0).sum())
    assert outside > 0 if dense else outside <= 0.02 * pairs          # surfaces fit their windows (almost) always


def reference_f64(x, w, nbr, scale, shift, res, relu):
    y = np.zeros((nbr.shape[0], w.shape[2]))
    for j in range(27):
        v = nbr[:, j] >= 0
        y[v] += x[nbr[v, j]].astype(np.float64) @ w[j].astype(np.float64)
    if scale is not None:
        y = y * scale + shift + res
    return np.maximum(y, 0) if relu else y


@pytest.mark.parametrize("cin,cout,n,ts,dense,batch", [(96, 96, 40000, 1, False, 1), (128, 96, 30000, 1, False, 1),
                                                        (32, 32, 40000, 2, False, 1), (64, 64, 20000, 1, False, 1),
                                                        (96, 96, 300, 1, False, 1), (32, 96, 17001, 1, False, 1),
                                                        (96, 96, 8000, 1, True, 1), (32, 32, 4096, 1, True, 1),
                                                        (96, 96, 15000, 1, False, 3), (96, 96, 300000, 1, False, 1)])
def test_conv_win_matches_conv_hl_and_float64(cuda, built_lib, cin, cout, n, ts, dense, batch):
    """conv_win against the mask-sorted kernels on the same hl operands (plain output; folded affine + hl residual + ReLU +
    hl output), ragged last tiles, a tile count below the XCD remap, and a solid block whose windows overflow (the pairs
    outside the window come through the extra units)."""
    cm = sorted_manager(cuda, scene(cin + cout, n, dense, batch))
    nbr = cm.kernel_map(3, ts)
    N = nbr.shape[0]
    rng = np.random.default_rng(cin + 3 * cout + n)
    t = lambda a: torch.from_numpy(a).to(cuda)
    x = rng.normal(0, 1, (N, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    res = rng.normal(0, 1, (N, cout)).astype(np.float32)
    xh, rh, wt = ME.to_hl(t(x)), ME.to_hl(t(res)), t(w)
    win = cm.windows(ts)

    def run(**kw):
        plain = ME.conv_forward(xh, wt, nbr, N, pieces=2, in_hl=True, **kw)
        out = torch.empty((N, cout), device=cuda)
        ME.conv_forward(xh, wt, nbr, N, pieces=2, in_hl=True, scale=t(scale), shift=t(shift), residual=rh, relu=True,
                        out=out, out_hl=True, res_hl=True, **kw)
        return plain, ME.from_hl(out)

    prev = ME.set_option("win", 1)
    try:
        got = run(win=win)
        ME.set_option("win", 0)
        want = run(win=win)              # the switch off: the same call takes the mask-sorted / split kernels
        ME.set_option("win", 1)
        again = run(win=win)
    finally:
        ME.set_option("win", prev)
    nb = nbr.cpu().numpy()
    # (the float64 gather-matmul of the 300k-point case would take minutes of numpy: there the mask-sorted kernels are the check)
    ref = (reference_f64(x, w, nb, None, None, None, False), reference_f64(x, w, nb, scale, shift, res, True)) if N <= 100000 \
        else (None, None)
    for g, h, r in zip(got, want, ref):
        tol = 1e-5 * max(1.0, float(h.abs().max()))
        assert float((g - h).abs().max()) < tol
        assert r is None or np.abs(g.cpu().numpy() - r).max() < tol
    assert float(got[0].abs().max()) > 0.1
    assert torch.equal(again[0], got[0]) and torch.equal(again[1], got[1])          # the same bits on every run


def test_program_forward_on_windows_matches_mask_sorted_program(cuda, built_lib):
    """The fused network with the fine levels on neighbour windows (option "win" on) against the same program with the option
    off (mask-sorted conv_hl / conv_hd + finish launches): per-point outputs agree within the north_star tolerance; the
    plan built windows for exactly the levels cv_net_win_levels names and no mask orders there."""
    from canonicalvoting_amd.minkunet import MinkUNet34C
    n = 40000
    sc = make_scene(5, n_points=n)
    c4 = torch.from_numpy(np.concatenate([np.zeros((n, 1), np.int64), sc.coords], 1)).to(cuda, torch.int32)
    f = torch.from_numpy((sc.feats * 2 - 1).astype(np.float32)).to(cuda)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).to(cuda).eval()
    prev = ME.set_option("win", 1)
    try:
        with torch.no_grad():
            x = ME.SparseTensor(f, c4, device=cuda)
            y_win = model.program_forward(x).F
            plan = x.coordinate_manager.fused_fast(5, 3)
            assert [p is not None for p in plan.win_ptrs] == [True, plan.counts[1] >= ME.CoordinateManager.MASKED_MIN_ROWS, False, False, False]
            assert plan.perm_ptrs[0] is None
            ME.set_option("win", 0)
            x2 = ME.SparseTensor(f, c4, device=cuda)
            y_old = model.program_forward(x2).F
            assert all(p is None for p in x2.coordinate_manager.fused_fast(5, 0).win_ptrs)
    finally:
        ME.set_option("win", prev)
    assert float((y_win - y_old).abs().max()) < 1e-4 * max(1.0, float(y_old.abs().max()))
    assert float(y_old.abs().max()) > 1e-3
