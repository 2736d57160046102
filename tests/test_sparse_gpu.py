"""gfx950 sparse-voxel engine vs oracle/sparse_oracle.py: coordinate sets and kernel maps exact
(integer work), convolution / network floats within the north_star tolerance 1e-4."""
import numpy as np
import pytest
import torch

from oracle import sparse_oracle as so
from canonicalvoting_amd import me as ME
from canonicalvoting_amd import pipeline
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene

pytestmark = pytest.mark.gpu


def scene_coords(seed, n, batch=1, small=True):
    cs, fs = [], []
    for b in range(batch):
        if small:
            sc = make_scene(seed + b, n_points=n, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5,
                            box_scale=0.4)
        else:
            sc = make_scene(seed + b, n_points=n)
        cs.append(np.concatenate([np.full((n, 1), b, np.int64), sc.coords], 1))
        fs.append(sc.feats * 2 - 1)
    return np.concatenate(cs), np.concatenate(fs).astype(np.float32)


@pytest.mark.parametrize("seed,n,batch", [(0, 700, 1), (1, 3000, 2)])
def test_coordinate_sets_and_kernel_maps_exact(cuda, built_lib, seed, n, batch):
    coords, _ = scene_coords(seed, n, batch)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    ocm = so.CoordinateManager(coords)
    for ts in (1, 2, 4, 8, 16):
        assert cm.num_rows(ts) == len(ocm.coords[ts])
        assert np.array_equal(cm.coords[ts].cpu().numpy(), ocm.coords[ts])       # same order too
    for k, ts, stride in ((5, 1, 1), (3, 1, 1), (3, 2, 1), (3, 4, 1), (3, 8, 1), (3, 16, 1), (2, 1, 2),
                          (2, 2, 2), (2, 4, 2), (2, 8, 2)):
        assert np.array_equal(cm.kernel_map(k, ts, stride).cpu().numpy(), ocm.map(k, ts, stride)), (k, ts, stride)
    for ts_c in (2, 4, 8, 16):
        up = cm.up_map(ts_c).cpu().numpy()
        down = ocm.map(2, ts_c // 2, 2)
        assert ((up >= 0).sum(1) == 1).all()                      # exactly one parent / octant per fine row
        f, j = np.nonzero(up >= 0)
        assert np.array_equal(down[up[f, j], j], f)


@pytest.mark.parametrize("seed,n,batch", [(3, 1, 1), (0, 700, 1), (1, 3000, 3), (2, 80000, 1)])
def test_spatial_row_sort_is_the_stable_sort_of_its_key(cuda, built_lib, seed, n, batch):
    """cv_sp_sort_rows (the row order the fused network runs on): key = batch | Z-order of the coarse cube,
    stable, so perm equals numpy's stable argsort of the same key bit for bit; inv is its inverse."""
    import ctypes
    from canonicalvoting_amd import _lib
    L = _lib.lib()
    coords, _ = scene_coords(seed, n, batch, small=n < 10000)
    rng = np.random.default_rng(seed)
    coords = coords[rng.permutation(len(coords))]             # scenes of the batch interleaved, rows shuffled
    coords[:, 1:] -= 7                                        # negative coordinates too
    N = len(coords)
    c = torch.from_numpy(coords).to(cuda, torch.int32).contiguous()
    srt = torch.empty((N, 4), dtype=torch.int32, device=cuda)
    perm = torch.empty(N, dtype=torch.int32, device=cuda)
    inv = torch.empty(N, dtype=torch.int32, device=cuda)
    ws = torch.empty(int(L.cv_sp_sort_workspace_bytes(N)), dtype=torch.uint8, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.cv_sp_sort_rows(c.data_ptr(), N, srt.data_ptr(), perm.data_ptr(), inv.data_ptr(), ws.data_ptr(),
                                 ws.numel(), st), "cv_sp_sort_rows")
    mn, mx = coords[:, 1:].min(0), coords[:, 1:].max(0)
    shift = 0
    while (int((mx - mn).max()) >> shift) >= 64:
        shift += 1
    q = ((coords[:, 1:] - mn) >> shift).astype(np.uint64)
    key = coords[:, 0].astype(np.uint64) << np.uint64(18)
    for bit in range(6):
        for ax in range(3):
            key |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    ref = np.argsort(key, kind="stable")
    assert np.array_equal(perm.cpu().numpy(), ref)
    assert np.array_equal(srt.cpu().numpy(), coords[ref])
    assert np.array_equal(inv.cpu().numpy()[ref], np.arange(N))


@pytest.mark.parametrize("n,batch", [(3000, 1), (2000, 2), (40000, 1)])
def test_scene_plan_in_one_call_equals_the_step_by_step_plan(cuda, built_lib, n, batch):
    """cv_sp_scene_plan (sort + levels + every map and order in one call, counts through pinned memory) against the
    same plan built call by call (cv_sp_sort_rows, cv_sp_build_levels, cv_sp_kernel_map / up maps of a
    CoordinateManager over the sorted rows): coordinate sets, kernel maps and transposed maps exact; the mask-sorted
    orders are permutations that group equal masks (their order inside a class is not defined)"""
    import ctypes
    from canonicalvoting_amd import _lib
    L = _lib.lib()
    coords, _ = scene_coords(7, n, batch, small=n < 10000)
    rng = np.random.default_rng(n)
    coords = coords[rng.permutation(len(coords))]
    N = len(coords)
    c = torch.from_numpy(coords).to(cuda, torch.int32).contiguous()
    cm = ME.CoordinateManager(c, num_levels=1, check=False, lazy=True)
    cm_s, stem_map, out_map = cm.fused_plan(5)
    # step by step
    srt = torch.empty((N, 4), dtype=torch.int32, device=cuda)
    perm = torch.empty(N, dtype=torch.int32, device=cuda)
    inv = torch.empty(N, dtype=torch.int32, device=cuda)
    ws = torch.empty(int(L.cv_sp_sort_workspace_bytes(N)), dtype=torch.uint8, device=cuda)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.cv_sp_sort_rows(c.data_ptr(), N, srt.data_ptr(), perm.data_ptr(), inv.data_ptr(), ws.data_ptr(),
                                 ws.numel(), st), "cv_sp_sort_rows")
    ref = ME.CoordinateManager(srt)
    assert torch.equal(out_map.view(-1), inv)
    for ts in (1, 2, 4, 8, 16):
        assert cm_s.num_rows(ts) == ref.num_rows(ts)
        assert torch.equal(cm_s.coords[ts], ref.coords[ts])
        assert torch.equal(cm_s.kernel_map(3, ts), ref.kernel_map(3, ts))
    for ts in (1, 2, 4, 8):
        assert torch.equal(cm_s.kernel_map(2, ts, 2), ref.kernel_map(2, ts, 2))
        assert torch.equal(cm_s.up_map(2 * ts), ref.up_map(2 * ts))
    # stem map: the sorted set's own 5x5x5 map with the sort permutation folded in
    k5 = ref.kernel_map(5, 1)
    expect = torch.where(k5 >= 0, perm[k5.clamp_min(0).long()], k5)
    assert torch.equal(stem_map, expect)
    G = ME.CoordinateManager.MASK_GROUPS
    for ts in (1, 2):
        if cm_s.num_rows(ts) >= ME.CoordinateManager.MASKED_MIN_ROWS:
            mp = cm_s.mask_perms(3, ts, G).cpu().numpy()
            nbr = cm_s.kernel_map(3, ts).cpu().numpy()
            for g in range(G):
                jb, je = 27 * g // G, 27 * (g + 1) // G
                assert np.array_equal(np.sort(mp[g]), np.arange(nbr.shape[0]))
                key = ((nbr[mp[g], jb:je] >= 0) * (1 << np.arange(je - jb))).sum(1)
                change = np.nonzero(np.diff(key))[0]
                assert len(np.unique(key)) == len(change) + 1           # every mask value is one contiguous run


def test_duplicate_coordinates_rejected(cuda, built_lib):
    c = torch.tensor([[0, 1, 2, 3], [0, 1, 2, 3], [0, 4, 5, 6]], dtype=torch.int32, device=cuda)
    with pytest.raises(RuntimeError, match="duplicate"):      # reported at the first map request
        ME.MinkowskiConvolution(3, 8, kernel_size=3, dimension=3).cuda()(ME.SparseTensor(torch.zeros(3, 3, device=cuda), c))
    model = MinkUNet34C(3, 8).cuda().eval()
    with pytest.raises(RuntimeError, match="duplicate"), torch.no_grad():
        model(ME.SparseTensor(torch.zeros(3, 3, device=cuda), c))


def test_coordinates_outside_the_key_window_rejected(cuda, built_lib):
    """the coordinate hash packs 16 bits per field: anything that could alias is refused, not mis-mapped"""
    for bad in ([0, 40000, 0, 0], [0, 0, -32768, 0], [70000, 1, 2, 3]):
        c = torch.tensor([[0, 1, 2, 3], bad], dtype=torch.int32, device=cuda)
        with pytest.raises(RuntimeError, match="outside the supported window"):
            ME.MinkowskiConvolution(3, 8, kernel_size=3, dimension=3).cuda()(ME.SparseTensor(torch.zeros(2, 3, device=cuda), c))
    ok = torch.tensor([[0, 32703, -32704, 3], [65535, 1, 2, 3]], dtype=torch.int32, device=cuda)
    ME.MinkowskiConvolution(3, 8, kernel_size=3, dimension=3).cuda()(ME.SparseTensor(torch.zeros(2, 3, device=cuda), ok))


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


@pytest.mark.parametrize("cin,cout,k,flavour", [(32, 32, 3, 1), (32, 64, 3, 1), (96, 96, 3, 1), (128, 96, 3, 1),
                                               (64, 128, 3, 1), (128, 256, 3, 1), (256, 256, 3, 0),
                                               (64, 64, 3, 0), (32, 32, 3, 0), (3, 32, 5, 1), (6, 32, 5, 1),
                                               (3, 32, 5, 0), (6, 32, 5, 0), (3, 32, 3, 0),   # flavour 0: stem kernel

                                               (128, 96, 1, 1), (96, 64, 1, 1), (384, 256, 1, 0)])
def test_conv_matches_oracle(cuda, built_lib, cin, cout, k, flavour):
    coords, _ = scene_coords(2, 1500)
    rng = np.random.default_rng(cin * 1000 + cout)
    x = rng.normal(0, 1, (len(coords), cin)).astype(np.float32)
    w = (rng.normal(0, 1, (k ** 3, cin, cout)) / np.sqrt(cin * k ** 3)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rng.normal(0, 0.2, cout).astype(np.float32)
    res = rng.normal(0, 1, (len(coords), cout)).astype(np.float32)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr = cm.kernel_map(k, 1) if k > 1 else None
    t = lambda a: torch.from_numpy(a).to(cuda)
    onbr = so.kernel_map(coords, coords, k, 1, 1)
    ref0 = so.conv(torch.from_numpy(x), torch.from_numpy(w), onbr).numpy()
    got0 = ME.conv_forward(t(x), t(w), nbr, len(coords), flavour=flavour).cpu().numpy()
    assert rel_err(got0, ref0) < 1e-5
    ref1 = np.maximum(ref0 * scale + shift + res, 0)
    # strided input / output / residual views (the fused network writes into concat buffers)
    xin = torch.zeros((len(coords), cin + 32), device=cuda); xin[:, 16:16 + cin] = t(x)
    rbuf = torch.zeros((len(coords), cout + 8), device=cuda); rbuf[:, 4:4 + cout] = t(res)
    obuf = torch.full((len(coords), cout + 64), -7.0, device=cuda)
    ME.conv_forward(xin[:, 16:16 + cin], t(w), nbr, len(coords), scale=t(scale), shift=t(shift),
                    residual=rbuf[:, 4:4 + cout], relu=True, out=obuf[:, 32:32 + cout], flavour=flavour)
    got1 = obuf[:, 32:32 + cout].cpu().numpy()
    assert rel_err(got1, ref1) < 1e-5
    assert float(obuf[:, :32].min()) == -7.0 and float(obuf[:, 32 + cout:].max()) == -7.0   # no stray writes


@pytest.mark.parametrize("cin,cout,k,n,masked", [(32, 32, 3, 1500, False), (64, 96, 3, 1500, False), (96, 64, 1, 1500, False),
                                                 (32, 256, 3, 13000, False),       # >= 384 tiles, not mask-sorted: forced 3-way split

                                                 (128, 128, 3, 600, False), (256, 256, 3, 300, False),
                                                 (32, 32, 2, 1500, False), (96, 96, 3, 20000, True),
                                                 (128, 96, 3, 20000, True)])
def test_hl_format_conv_matches_fp32_operand_conv(cuda, built_lib, cin, cout, k, n, masked):
    """activations in the hl format (the fp16 pair of every value stored in place, cv_conv_desc.in_hl): the format
    round-trips to 2^-24, a convolution reading it has the SAME accumulators as the fp16-pair kernel splitting fp32
    rows on the fly (bit-identical fp32 outputs), and the hl epilogue (residual read, output split) stays at fp32 level"""
    coords, _ = scene_coords(2, n, small=n < 10000)
    N = len(coords)
    rng = np.random.default_rng(cin * 7 + cout)
    t = lambda a: torch.from_numpy(a).to(cuda)
    x = t(rng.normal(0, 1, (N, cin)).astype(np.float32))
    x[::7] *= 1e-3
    x[::11] *= 300.0
    w = t((rng.normal(0, 1, (k ** 3, cin, cout)) / np.sqrt(cin * k ** 3)).astype(np.float32))
    scale = t(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = t(rng.normal(0, 0.2, cout).astype(np.float32))
    res = t(rng.normal(0, 1, (N, cout)).astype(np.float32))
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    if k == 2:
        nbr, n_out = cm.kernel_map(2, 1, 2), cm.num_rows(2)
        res = res[:n_out].contiguous()
    else:
        nbr, n_out = (cm.kernel_map(k, 1) if k > 1 else None), N
    xh = ME.to_hl(x)
    back = ME.from_hl(xh)
    assert bool(((back - x).abs() <= x.abs() * 2.0 ** -23 + 2.0 ** -24).all())     # low pieces below 2^-14 are fp16 subnormals
    kw = dict(pieces=2)
    if masked:
        perms = cm.mask_perms(3, 1, 4)
        conv = lambda xin, **e: ME.conv_forward_masked(xin, w, nbr, perms, n_out, **kw, **e)
    else:
        conv = lambda xin, **e: ME.conv_forward(xin, w, nbr, n_out, **kw, **e)
    ref0 = conv(x)
    got0 = conv(xh, in_hl=True)
    if n == 13000:      # the hl launch is split three ways (conv_hl holds <= 10 offsets per workgroup), the fp32 one is not
        assert rel_err(got0.cpu().numpy(), ref0.cpu().numpy()) < 1e-6
    else:
        assert torch.equal(got0, ref0)
    ref1 = conv(x, scale=scale, shift=shift, residual=res, relu=True)
    if not masked:
        # split-K reduced by the last-arriving workgroup of every output tile (cv_conv_desc.split_tickets) instead of a
        # finish launch: same summation order, bit-identical; the counters are left at zero
        tickets = torch.zeros(4096, dtype=torch.int32, device=cuda)
        for _ in range(2):
            got_f = conv(xh, in_hl=True, scale=scale, shift=shift, residual=res, relu=True, split_tickets=tickets)
            got_u = conv(xh, in_hl=True, scale=scale, shift=shift, residual=res, relu=True)
            assert torch.equal(got_f, got_u)
            assert int(tickets.abs().sum()) == 0
    # hl in / residual / out inside wider buffers (column windows at multiples of 32 channels)
    xin = torch.zeros((N, cin + 64), device=cuda); xin[:, 32:32 + cin] = x
    xin_h = ME.to_hl(xin)
    rbuf = torch.zeros((n_out, cout + 32), device=cuda); rbuf[:, 32:] = res
    rbuf_h = ME.to_hl(rbuf)
    obuf = ME.to_hl(torch.full((n_out, cout + 64), -7.0, device=cuda))
    ME.conv_forward(xin_h[:, 32:32 + cin], w, nbr, n_out, scale=scale, shift=shift, residual=rbuf_h[:, 32:], relu=True,
                    out=obuf[:, 32:32 + cout], in_hl=True, out_hl=True, res_hl=True,
                    **(dict(row_perm=perms, perm_groups=4, pieces=2) if masked else kw))
    got1 = ME.from_hl(obuf)
    assert rel_err(got1[:, 32:32 + cout].cpu().numpy(), ref1.cpu().numpy()) < 1e-6
    assert float(got1[:, :32].min()) == -7.0 and float(got1[:, :32].max()) == -7.0       # no stray writes
    assert float(got1[:, 32 + cout:].min()) == -7.0 and float(got1[:, 32 + cout:].max()) == -7.0
    assert int(ME.range_flag(cuda)[0]) == 0
    # an output beyond the fp16 range raises the flag (the caller then falls back to fp32 buffers)
    ME.conv_forward(xh, w * 1e6, nbr, n_out, out=obuf[:, 32:32 + cout], in_hl=True, out_hl=True, **kw)
    torch.cuda.synchronize()
    assert int(ME.range_flag(cuda)[0]) == 1
    ME.range_flag(cuda).zero_()


@pytest.mark.parametrize("cin,n,hl", [(3, 1500, False), (3, 20000, True), (6, 1500, True)])
def test_matrix_core_stem_matches_oracle(cuda, built_lib, cin, n, hl):
    """5x5x5 stem as a GEMM over the gathered operand on the matrix cores (conv_stem_mfma, fp16 pairs, BatchNorm scale
    folded into the packed weights) against the oracle's fp32 convolution, fp32 and hl-format output"""
    coords, _ = scene_coords(4, n, small=n < 10000)
    N = len(coords)
    rng = np.random.default_rng(cin + n)
    t = lambda a: torch.from_numpy(a).to(cuda)
    x = rng.uniform(-1, 1, (N, cin)).astype(np.float32)
    w = (rng.normal(0, 1, (125, cin, 32)) / np.sqrt(cin * 16)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    shift = rng.normal(0, 0.2, 32).astype(np.float32)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr = cm.kernel_map(5, 1)
    ref = so.conv(torch.from_numpy(x), torch.from_numpy(w), so.kernel_map(coords, coords, 5, 1, 1)).numpy()
    ref = np.maximum(ref * scale + shift, 0)
    out = torch.full((N, 64), -7.0, device=cuda)
    if hl:
        out = ME.to_hl(out)
    ME.conv_forward(t(x), t(w), nbr, N, scale=t(scale), shift=t(shift), relu=True, out=out[:, 32:], pieces=2,
                    stem_mfma=True, out_hl=hl)
    got = (ME.from_hl(out) if hl else out).cpu().numpy()
    assert rel_err(got[:, 32:], ref) < 2e-6
    assert float(got[:, :32].min()) == -7.0 and float(got[:, :32].max()) == -7.0
    assert int(ME.range_flag(cuda)[0]) == 0
    # an input beyond the fp16 range raises the flag
    xb = t(x).clone(); xb[5, 0] = 1e5
    ME.conv_forward(xb, t(w), nbr, N, out=out[:, 32:], pieces=2, stem_mfma=True, out_hl=hl)
    torch.cuda.synchronize()
    assert int(ME.range_flag(cuda)[0]) == 1
    ME.range_flag(cuda).zero_()


def test_masked_two_pass_conv_matches_oracle(cuda, built_lib):
    """mask-sorted processing order + offset halves + acc_in give the same conv"""
    coords, _ = scene_coords(6, 5000, small=False)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 1, (len(coords), 64)).astype(np.float32)
    w = (rng.normal(0, 1, (27, 64, 96)) / 40).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 96).astype(np.float32)
    shift = rng.normal(0, 0.2, 96).astype(np.float32)
    res = rng.normal(0, 1, (len(coords), 96)).astype(np.float32)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    t = lambda a: torch.from_numpy(a).to(cuda)
    ref = so.conv(torch.from_numpy(x), torch.from_numpy(w), so.kernel_map(coords, coords, 3, 1, 1)).numpy()
    ref = np.maximum(ref * scale + shift + res, 0)
    for groups in (2, 3, 4):
        perms = cm.mask_perms(3, 1, groups)
        for p in perms:
            assert sorted(p.cpu().tolist()) == list(range(len(coords)))
        got = ME.conv_forward_masked(t(x), t(w), cm.kernel_map(3, 1), perms, len(coords), scale=t(scale),
                                     shift=t(shift), residual=t(res), relu=True).cpu().numpy()
        assert rel_err(got, ref) < 1e-5, groups
    for cout in (32, 64, 128, 256):            # every column-block width
        w2 = (rng.normal(0, 1, (27, 64, cout)) / 40).astype(np.float32)
        ref2 = so.conv(torch.from_numpy(x), torch.from_numpy(w2), so.kernel_map(coords, coords, 3, 1, 1)).numpy()
        got2 = ME.conv_forward_masked(t(x), t(w2), cm.kernel_map(3, 1), cm.mask_perms(3, 1, 3), len(coords)).cpu().numpy()
        assert rel_err(got2, ref2) < 1e-5, cout
    # explicit two-launch form: offset halves chained through acc_in
    perms = cm.mask_perms(3, 1, 2)
    part = ME.conv_forward(t(x), t(w), cm.kernel_map(3, 1), len(coords), flavour=1, row_perm=perms[0], j_begin=0, j_end=13)
    got = ME.conv_forward(t(x), t(w), cm.kernel_map(3, 1), len(coords), flavour=1, row_perm=perms[1], j_begin=13,
                          j_end=27, acc_in=part, scale=t(scale), shift=t(shift), residual=t(res), relu=True).cpu().numpy()
    assert rel_err(got, ref) < 1e-5


def test_strided_and_transposed_conv_match_oracle(cuda, built_lib):
    coords, _ = scene_coords(3, 2000)
    rng = np.random.default_rng(0)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    ocm = so.CoordinateManager(coords)
    x = rng.normal(0, 1, (len(coords), 32)).astype(np.float32)
    w = rng.normal(0, 0.1, (8, 32, 64)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(cuda)
    down = ME.conv_forward(t(x), t(w), cm.kernel_map(2, 1, 2), cm.num_rows(2))
    rdown = so.conv(torch.from_numpy(x), torch.from_numpy(w), ocm.map(2, 1, 2))
    assert rel_err(down.cpu().numpy(), rdown.numpy()) < 1e-5
    wt = rng.normal(0, 0.1, (8, 64, 96)).astype(np.float32)
    up = ME.conv_forward(down, t(wt), cm.up_map(2), cm.num_rows(1))
    rup = so.conv_transpose_k2s2(rdown, torch.from_numpy(wt), ocm.map(2, 1, 2))
    assert rel_err(up.cpu().numpy(), rup.numpy()) < 1e-5


def test_modules_match_oracle_and_reference_api(cuda, built_lib):
    """facade modules one by one (the unfused path the reference takes)"""
    coords, feats = scene_coords(4, 900)
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    assert x.F.shape == (900, 3) and x.C.shape == (900, 4)
    conv = ME.MinkowskiConvolution(3, 32, kernel_size=5, dimension=3).cuda()
    bn = ME.MinkowskiBatchNorm(32).cuda().eval()
    bn.bn.running_mean.normal_(0, 0.1); bn.bn.running_var.uniform_(0.5, 2)
    relu = ME.MinkowskiReLU(inplace=True)
    with torch.no_grad():
        y = relu(bn(conv(x)))
    sd = {"c.kernel": conv.kernel.detach().cpu()}
    sd.update({"b.bn." + k: v.detach().cpu() for k, v in bn.bn.state_dict().items()})
    ref = torch.relu(so.batch_norm(so.conv(torch.from_numpy(feats), sd["c.kernel"],
                                           so.kernel_map(coords, coords, 5, 1, 1)), sd, "b"))
    assert rel_err(y.F.cpu().numpy(), ref.numpy()) < 1e-5
    z = ME.cat(y, y)
    assert z.F.shape == (900, 64)
    with pytest.raises(RuntimeError, match="GPU only"):
        ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cpu")


@pytest.mark.parametrize("n,small", [(800, True), (8000, False)])
def test_minkunet34c_fused_and_modular_match_oracle(cuda, built_lib, n, small):
    coords, feats = scene_coords(5, n, small=small)
    sd = so.make_state_dict(3, 64, seed=1)
    model = MinkUNet34C(3, 64)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        fused = model(x).F.cpu().numpy()
        modular = model.modular_forward(x).F.cpu().numpy()
        program = model.program_forward(x).F.cpu().numpy()        # same launches through the C executor
    # the C program folds the 1x1 downsample branches into conv2 (BatchNorm scales folded into the packed weights):
    # same math, different rounding - equal to the Python-issued launches within fp32 noise, identical without the fusion
    assert np.abs(program - fused).max() < 2e-5 * max(1.0, np.abs(fused).max())
    saved = model.FUSE_DOWNSAMPLE
    try:
        type(model).FUSE_DOWNSAMPLE = False
        model.__dict__.pop("_prog", None)
        with torch.no_grad():
            assert np.array_equal(model.program_forward(x, pieces=3).F.cpu().numpy(), model.fused_forward(x).F.cpu().numpy())
    finally:
        type(model).FUSE_DOWNSAMPLE = saved
        model.__dict__.pop("_prog", None)
    ref = so.minkunet34c_forward(sd, coords, feats).numpy()
    assert fused.shape == ref.shape == (n, 64)
    # north_star: within 1e-4 on the LCC / scale floats (outputs are O(1))
    assert np.abs(fused - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    assert np.abs(modular - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # head split (eval_joint.py:173-190): integer outputs exact, floats 1e-4
    xyz, scale, prob, cls = pipeline.head_joint(torch.from_numpy(ref).to(cuda))
    rx, rs, rp, rc = so.head_joint_eval(ref)
    assert np.array_equal(cls.cpu().numpy(), rc.numpy())
    np.testing.assert_allclose(xyz.cpu().numpy(), rx.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(scale.cpu().numpy(), rs.numpy(), rtol=1e-5)
    np.testing.assert_allclose(prob.cpu().numpy(), rp.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("target", [64, 256, 2048])
def test_split_target_changes_launch_sizes_not_results(cuda, built_lib, target):
    """cv_sp_set_split_target (the launch sizing a host picks by its scenes in flight): the network output stays within
    the 1e-4 bar of the oracle for small / bench / large targets, the setter returns the previous value and 0 restores
    the default."""
    coords, feats = scene_coords(5, 8000, small=False)
    sd = so.make_state_dict(3, 64, seed=1)
    model = MinkUNet34C(3, 64)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    ref = so.minkunet34c_forward(sd, coords, feats).numpy()
    default = ME.set_split_target(target)
    try:
        assert ME.set_split_target(target) == target
        x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
        with torch.no_grad():
            out = model(x).F.cpu().numpy()
    finally:
        ME.set_split_target(0)
    assert ME.set_split_target(0) == default
    assert np.abs(out - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_minkunet_batch_of_scenes_and_row_order(cuda, built_lib):
    coords, feats = scene_coords(7, 1200, batch=3)
    sd = so.make_state_dict(3, 8, seed=2)                       # separate-model head (8 channels)
    model = MinkUNet34C(3, 8)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    perm = np.random.default_rng(0).permutation(len(coords))
    with torch.no_grad():
        y = model(ME.SparseTensor(torch.from_numpy(feats[perm]), torch.from_numpy(coords[perm]).int(),
                                  device="cuda")).F.cpu().numpy()
    ref = so.minkunet34c_forward(sd, coords, feats).numpy()
    assert np.abs(y - ref[perm]).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_detect_scene_end_to_end(cuda, built_lib):
    from canonicalvoting_amd.hough import HoughVoting
    coords, feats = scene_coords(9, 8000, small=False)
    model = MinkUNet34C(3, 64)
    model.load_state_dict(so.make_state_dict(3, 64, seed=3))
    model = model.cuda().eval()
    hv = HoughVoting(0.03, 120)
    dets, raw, y = pipeline.detect_scene(model, hv, torch.from_numpy(coords).int().to(cuda),
                                         torch.from_numpy(feats).to(cuda), 0.03, thresh_high=5.0)
    assert y.F.shape == (8000, 64) and torch.isfinite(y.F).all()
    assert len(raw["verdict"]) == len(raw["cand_idx"])


def test_separate_models_share_one_coordinate_manager(cuda, built_lib):
    """eval_separate.py path: several 8-channel models on one SparseTensor; head vs oracle, maps built once"""
    from canonicalvoting_amd.hough import HoughVoting
    coords, feats = scene_coords(15, 3000, small=False)
    sds = {c: so.make_state_dict(3, 8, seed=30 + i) for i, c in enumerate(("chair", "table"))}
    models = {}
    for c, sd in sds.items():
        m = MinkUNet34C(3, 8)
        m.load_state_dict(sd)
        models[c] = m.cuda().eval()
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        y = {c: m(x).F for c, m in models.items()}
    plan = x.coordinate_manager.fused_plan()
    assert x.coordinate_manager.fused_plan() is plan                      # cached: built once for all models
    for c in models:
        ref = so.minkunet34c_forward(sds[c], coords, feats)
        assert np.abs(y[c].cpu().numpy() - ref.numpy()).max() < 1e-4 * max(1.0, float(ref.abs().max()))
        xyz, scale, prob = pipeline.head_separate(y[c])
        rx, rs, rp = so.head_separate_eval(y[c].cpu())
        np.testing.assert_allclose(xyz.cpu().numpy(), rx.numpy(), rtol=0, atol=0)
        np.testing.assert_allclose(scale.cpu().numpy(), rs.numpy(), rtol=1e-5)
        np.testing.assert_allclose(prob.cpu().numpy(), rp.numpy(), rtol=1e-5, atol=1e-6)
    dets = pipeline.detect_scene_separate(models, HoughVoting(0.03, 120), torch.from_numpy(coords).int().to(cuda),
                                          torch.from_numpy(feats).to(cuda), 0.03, thresh_high=3.0)
    assert all(d[0] in models for d in dets)


def test_eval_script_pipeline_recovers_planted_boxes(cuda, built_lib):
    """eval_joint.py-shaped loop on synthetic scans with teacher-forced predictions: network forward runs,
    vote + decode + NMS recover the planted boxes and the mAP evaluator scores them (end-to-end sanity of
    vote -> decode -> NMS -> calc_map against ground truth that none of those stages has seen)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "cv_eval_joint", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts", "eval_joint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from canonicalvoting_amd.data import SyntheticScanDataset, collate_fn
    ds = SyntheticScanDataset(2, 40000, seed0=200)
    batch = collate_fn([ds[0], ds[1]])
    assert batch[1].shape == (80000, 4) and batch[1].dtype == torch.int32 and int(batch[1][:, 0].max()) == 1
    assert batch[5].dtype == torch.int64 and len(ds.gt_lines(0)) == 12
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().eval()
    res = mod.evaluate(model, ds, teacher=True)
    assert res[0.25]["mAP"] > 0.85 and res[0.25]["AR"] > 0.85, res
    assert res[0.5]["mAP"] > 0.5, res


def test_real_file_dataset_through_eval_and_train_step(cuda, built_lib, tmp_path):
    """ScanNet/Scan2CAD-format files (tests/golden/scannet_mini) -> reader -> collate -> eval loop and one training
    step: the real data path reaches the HIP network with the shapes/dtypes the synthetic one has."""
    import importlib.util
    import os
    from canonicalvoting_amd import data, train
    from tests.golden.make_data_golden import mini_cfg
    spec = importlib.util.spec_from_file_location(
        "cv_eval_joint", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts", "eval_joint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mini_cfg()
    cfg.data.gt_path = str(tmp_path)
    for sid in ("scene0000_00", "scene0001_00"):
        (tmp_path / (sid + ".txt")).write_text("0.5 0.4 -1.0 0.3 0.4 0.5 0.6 03001627\n")
    ds = data.ScanNetXYZProbMultiDataset(cfg, training=False, augment=False)
    torch.manual_seed(0)
    model = MinkUNet34C(3, 64).cuda().eval()
    res = mod.evaluate(model, ds, res=cfg.scannet_res)
    assert set(res) == {0.25, 0.5} and 0.0 <= res[0.25]["mAP"] <= 1.0
    _, coords, feats, xyz, scale, cls = data.collate_fn([ds[0], ds[1]])
    model.train()
    opt = train.make_optimizer(model, lr=1e-3)
    loss, _ = train.train_step(model, opt, coords.to(cuda), feats.to(cuda) * 2 - 1, xyz.to(cuda), scale.to(cuda), cls.to(cuda))
    assert np.isfinite(float(loss))
    # the per-category model of train_separate.py on the symmetric dataset: the loss falls over a few steps
    ds = data.ScanNetXYZProbSymDataset(mini_cfg(category="03001627"), training=False, augment=False)
    _, coords, feats, xyz_l, scale_l, obj_l, _ = data.collate_fn_separate([ds[0], ds[1]])
    torch.manual_seed(1)
    sep = MinkUNet34C(3, 8).cuda().train()
    opt = train.make_optimizer(sep, lr=1e-3)
    hist = [float(train.train_step_separate(sep, opt, coords.to(cuda), feats.to(cuda) * 2 - 1, xyz_l, scale_l, obj_l)[0])
            for _ in range(6)]
    assert np.isfinite(hist).all() and hist[-1] < hist[0], hist
    assert train.train_step_separate(sep, opt, coords.to(cuda), feats.to(cuda), xyz_l, scale_l, torch.zeros_like(obj_l)) is None


def test_scenes_in_flight_on_separate_streams_match_sequential(cuda, built_lib):
    """bench.py keeps several scenes in flight (one host thread + HIP stream each); per-scene results must be the
    ones of the one-at-a-time path, bit for bit (per-stream workspaces, no shared mutable state)."""
    import threading
    from canonicalvoting_amd.hough import HoughVoting
    scenes = []
    for seed in (21, 22, 23):
        coords, feats = scene_coords(seed, 6000, small=False)
        scenes.append((torch.from_numpy(coords).int().to(cuda), torch.from_numpy(feats).to(cuda)))
    model = MinkUNet34C(3, 64)
    model.load_state_dict(so.make_state_dict(3, 64, seed=4))
    model = model.cuda().eval()

    def run(i, hv, out):
        dets, raw, y = pipeline.detect_scene(model, hv, scenes[i][0], scenes[i][1], 0.03, thresh_high=5.0)
        out[i] = (y.F.clone(), np.asarray(raw["cand_idx"]).copy(), np.asarray(raw["verdict"]).copy(),
                  np.asarray(raw["boxes"]).copy())

    seq = {}
    for i in range(3):
        run(i, HoughVoting(0.03, 120), seq)
    torch.cuda.synchronize()
    par = {}
    streams = [torch.cuda.Stream(cuda) for _ in range(3)]

    def worker(i):
        torch.cuda.set_device(cuda)
        with torch.cuda.stream(streams[i]):
            for _ in range(3):                          # repeated: later rounds overlap fully
                run(i, HoughVoting(0.03, 120), par)
            streams[i].synchronize()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(3):
        assert torch.equal(seq[i][0], par[i][0])
        for a, b in zip(seq[i][1:], par[i][1:]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("case", ["single_voxel", "ragged_batch", "extreme_coordinates", "float_coordinates"])
def test_minkunet_edge_case_coordinate_sets(cuda, built_lib, case):
    """edge cases of the coordinate manager through the whole network: one voxel, a batch of very unequal scenes,
    coordinates near the +-32767 key range with negative values crossing the stride-16 floor division, float
    coordinates (floored like ME.SparseTensor does, train_joint.py:250)"""
    rng = np.random.default_rng(3)
    if case == "single_voxel":
        coords = np.array([[0, 5, -3, 7]], np.int64)
    elif case == "ragged_batch":
        a, _ = scene_coords(31, 1500)
        b = np.array([[1, 0, 0, 0], [1, 1, 0, 0], [1, 40, -7, 3]], np.int64)
        coords = np.concatenate([a, b])
    elif case == "extreme_coordinates":
        base, _ = scene_coords(32, 800)
        lo, hi = base.copy(), base.copy()
        lo[:, 1:] += np.array([-32700, -32000, -31000]) - base[:, 1:].min(0)
        hi[:, 1:] += np.array([32700, 32000, 31000]) - base[:, 1:].max(0)
        hi[:, 0] = 1
        coords = np.concatenate([lo, hi])
    else:
        coords, _ = scene_coords(33, 600)
    feats = rng.normal(0, 1, (len(coords), 3)).astype(np.float32)
    sd = so.make_state_dict(3, 8, seed=7)
    model = MinkUNet34C(3, 8)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    tc = torch.from_numpy(coords)
    if case == "float_coordinates":
        tc = tc.float() + torch.from_numpy(rng.uniform(0.0, 0.9, coords.shape).astype(np.float32))
        tc[:, 0] = torch.from_numpy(coords[:, 0]).float()
    else:
        tc = tc.int()
    with torch.no_grad():
        x = ME.SparseTensor(torch.from_numpy(feats), tc, device="cuda")
        y = model(x).F.cpu().numpy()
        ym = model.modular_forward(x).F.cpu().numpy()
    ref = so.minkunet34c_forward(sd, coords, feats).numpy()
    tol = 1e-4 * max(1.0, np.abs(ref).max())
    assert y.shape == ref.shape and np.abs(y - ref).max() < tol and np.abs(ym - ref).max() < tol


def test_bf16x6_products_keep_fp32_accuracy(cuda, built_lib):
    """conv_rows_wp (bf16 triples) computes every fp32 product as six exact bf16 piece products: against a float64 reference its
    error must stay at fp32-rounding level, next to the fp32-MFMA kernel's own error on the same inputs (operands with
    a wide dynamic range, so the low pieces matter)."""
    coords, _ = scene_coords(41, 3000, small=False)
    rng = np.random.default_rng(9)
    n = len(coords)
    x = (rng.normal(0, 1, (n, 96)) * np.exp(rng.normal(0, 2, (n, 96)))).astype(np.float32)
    w = (rng.normal(0, 1, (27, 96, 128)) * np.exp(rng.normal(0, 1.5, (27, 96, 128))) / 50).astype(np.float32)
    onbr = so.kernel_map(coords, coords, 3, 1, 1)
    ref = np.zeros((n, 128), np.float64)
    mag = np.zeros((n, 128), np.float64)
    for j in range(27):
        sel = np.nonzero(onbr[:, j] >= 0)[0]
        ref[sel] += x[onbr[sel, j]].astype(np.float64) @ w[j].astype(np.float64)
        mag[sel] += np.abs(x[onbr[sel, j]]).astype(np.float64) @ np.abs(w[j]).astype(np.float64)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    t = lambda a: torch.from_numpy(a).to(cuda)
    errs = {}
    saved = ME.CONV_X6
    try:
        for mode in (True, False):
            ME.CONV_X6 = mode
            got = ME.conv_forward(t(x), t(w), cm.kernel_map(3, 1), n, flavour=1).cpu().numpy().astype(np.float64)
            errs[mode] = float((np.abs(got - ref) / (mag + 1e-30)).max())     # error relative to sum |x||w|
    finally:
        ME.CONV_X6 = saved
    assert errs[False] < 2e-6 and errs[True] < 2e-6, errs                    # both at fp32 accumulation level
    assert errs[True] < 4 * errs[False] + 1e-7, errs
    # fp16 pairs (three piece products, weights pre-scaled by a power of two): the same bar, on the plain and the
    # mask-grouped launch; no input leaves the fp16 range here, so the flag stays down
    flag = ME.range_flag(torch.device(cuda))
    flag.zero_()
    for kw in (dict(flavour=1), dict()):
        got = ME.conv_forward(t(x), t(w), cm.kernel_map(3, 1), n, pieces=2, **kw).cpu().numpy().astype(np.float64)
        e2 = float((np.abs(got - ref) / (mag + 1e-30)).max())
        assert e2 < 2e-6 and e2 < 4 * errs[False] + 1e-7, (e2, errs)
    assert int(flag[0]) == 0
    # tiny and huge magnitudes: subnormal low pieces cost absolute, not relative accuracy; 7e4 raises the flag
    xs = x.copy()
    xs[:, :8] *= 1e-6
    got = ME.conv_forward(t(xs), t(w), cm.kernel_map(3, 1), n, pieces=2, flavour=1).cpu().numpy().astype(np.float64)
    want = ME.conv_forward(t(xs), t(w), cm.kernel_map(3, 1), n, flavour=1).cpu().numpy().astype(np.float64)
    assert float((np.abs(got - want) / (mag + 1e-30)).max()) < 2e-6
    assert int(flag[0]) == 0
    xs[5, 3] = 7e4
    ME.conv_forward(t(xs), t(w), cm.kernel_map(3, 1), n, pieces=2, flavour=1)
    torch.cuda.synchronize()
    assert int(flag[0]) == 1
    flag.zero_()


def test_fp16_range_overflow_falls_back_to_bf16_triples(cuda, built_lib):
    """An activation beyond the fp16 range (forced here by a BatchNorm gain of 3e6) must not corrupt the output: the
    flag goes up, the forward is redone on the bf16 triples, and the result equals the bf16 program's bit for bit."""
    coords, feats = scene_coords(43, 1500)
    torch.manual_seed(3)
    model = MinkUNet34C(3, 64).cuda().eval()
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        calm = model(x).F.clone()
        assert getattr(model, "range_fallbacks", 0) == 0
        assert float((calm - model.program_forward(x, pieces=3).F).abs().max()) < 2e-5 * max(1.0, float(calm.abs().max()))
        model.bn0.bn.weight.mul_(3e6)
        model.train(False)                                   # drops the folded affines and the programs
        want = model.program_forward(x, pieces=3).F.clone()
        got = model(x).F
        assert model.range_fallbacks == 1 and torch.equal(got, want)
        # the deferred form used by the per-scene pipeline
        y = model(x, defer_check=True)
        torch.cuda.synchronize()
        y2 = model.check_range(x, y)
        assert y2 is not y and torch.equal(y2.F, want) and model.range_fallbacks == 2


def test_hip_network_matches_reference_class_executed_on_cpu(cuda, built_lib):
    """tests/golden/net_ref.npz: the reference's own MinkUNet34C module tree and forward (utils/minkunet.py:36-180,
    utils/resnet.py:118-154) executed on CPU over the oracle's primitive ops (tests/golden/make_net_golden.py).
    The HIP network (C program, Python-issued fused launches, module-by-module) loads the same state dict and must
    give the same per-point outputs within north_star's 1e-4, eval and training-mode BatchNorm."""
    import json, os
    from tests.golden.make_net_golden import make_inputs
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "net_ref.npz"))
    coords, feats = make_inputs()
    model = MinkUNet34C(3, 64)
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == json.loads(str(z["state_dict"]))
    model.load_state_dict(so.make_state_dict(3, 64, seed=int(z["seed_w"])))
    model = model.cuda().eval()
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    tol = 1e-4 * max(1.0, np.abs(z["out_eval"]).max())
    with torch.no_grad():
        for fwd in (model.program_forward, model.fused_forward, model.modular_forward):
            assert np.abs(fwd(x).F.cpu().numpy() - z["out_eval"]).max() < tol
        model.train()
        y = model(x).F.cpu().numpy()
    assert np.abs(y - z["out_train"]).max() < 1e-4 * max(1.0, np.abs(z["out_train"]).max())


@pytest.mark.parametrize("wrapper", [None, "model_state_dict"])
def test_reference_checkpoint_in_z_fastest_offset_order_loads_and_matches_oracle(cuda, built_lib, tmp_path, wrapper):
    """SURVEY 8 f-3 (eval_joint.py:152 torch.load + load_state_dict; sunrgbd/brnetcanon.py:167 the
    ``model_state_dict`` wrapper): a checkpoint file whose [K^3, Cin, Cout] kernels enumerate the offsets with the LAST
    spatial axis fastest, written to disk, goes through load_reference_checkpoint(offset_order="z_fastest") into the
    HIP network; its output equals the oracle's run with the same file's weights under a z-fastest offset table, and
    the oracle's run with the un-permuted (x-fastest) weights.  Loading it as if it were x-fastest must NOT match."""
    from canonicalvoting_amd.minkunet import convert_kernel_offset_order, load_reference_checkpoint
    coords, feats = scene_coords(31, 1500, small=True)
    sd_x = so.make_state_dict(3, 64, seed=7)                                   # this engine's order
    sd_z = convert_kernel_offset_order(sd_x, "x_fastest", "z_fastest")          # what such a checkpoint would hold
    assert not torch.equal(sd_z["conv0p1s1.kernel"], sd_x["conv0p1s1.kernel"])
    path = str(tmp_path / "joint.pth")
    torch.save({wrapper: sd_z, "epoch": 3} if wrapper else sd_z, path)
    model = load_reference_checkpoint(MinkUNet34C(3, 64), path, offset_order="z_fastest", key=wrapper).cuda().eval()
    x = ME.SparseTensor(torch.from_numpy(feats), torch.from_numpy(coords).int(), device="cuda")
    with torch.no_grad():
        y = model(x).F.cpu().numpy()
    ref = so.minkunet34c_forward(sd_x, coords, feats).numpy()
    assert np.abs(y - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    old = so.KERNEL_OFFSET_ORDER
    try:
        so.KERNEL_OFFSET_ORDER = "z_fastest"                                    # the file's weights as they are
        ref_z = so.minkunet34c_forward(sd_z, coords, feats).numpy()
    finally:
        so.KERNEL_OFFSET_ORDER = old
    assert np.abs(y - ref_z).max() < 1e-4 * max(1.0, np.abs(ref_z).max())
    wrong = load_reference_checkpoint(MinkUNet34C(3, 64), path, offset_order="x_fastest", key=wrapper).cuda().eval()
    with torch.no_grad():
        assert np.abs(wrong(x).F.cpu().numpy() - ref).max() > 1e-2 * np.abs(ref).max()


@pytest.mark.parametrize("cin,cout,n,masked", [(96, 96, 40000, True), (128, 96, 40000, True), (32, 32, 20000, True),
                                                (64, 64, 20000, False), (96, 96, 300, False), (32, 64, 17000, True)])
def test_conv_hd_is_bit_identical_to_conv_hl(cuda, built_lib, cin, cout, n, masked):
    """conv_hd (LDS-DMA operand rings, 256-row workgroups, cv_sp_set_option "hd_mask") keeps conv_hl's units and MFMA
    sequence per accumulator: the same bits for mask-grouped and split launches, with a residual / ReLU / hl epilogue and a
    ragged last tile (n is no multiple of 256); the second source rides through the network-level comparison."""
    coords, _ = scene_coords(3, n, small=n < 10000)
    N = len(coords)
    rng = np.random.default_rng(cin + 3 * cout)
    t = lambda a: torch.from_numpy(a).to(cuda)
    x = t(rng.normal(0, 1, (N, cin)).astype(np.float32))
    w = t((rng.normal(0, 1, (27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32))
    scale = t(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    shift = t(rng.normal(0, 0.2, cout).astype(np.float32))
    res = t(rng.normal(0, 1, (N, cout)).astype(np.float32))
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr = cm.kernel_map(3, 1)
    xh, rh = ME.to_hl(x), ME.to_hl(res)
    if masked:
        perms = cm.mask_perms(3, 1, 3)
        conv = lambda **e: ME.conv_forward_masked(xh, w, nbr, perms, N, pieces=2, in_hl=True, **e)
    else:
        conv = lambda **e: ME.conv_forward(xh, w, nbr, N, pieces=2, in_hl=True, **e)

    def run():
        plain = conv()
        out = torch.empty((N, cout), device=cuda)
        conv(scale=scale, shift=shift, residual=rh, relu=True, out=out, out_hl=True, res_hl=True)
        return plain, out

    prev = ME.set_option("hd_mask", 0)
    prev_rows = ME.set_option("hd_min_rows", 1)
    try:
        want = run()
        ME.set_option("hd_mask", 7)
        got = []
        for shape in (0, 1, 2, 0):      # 8 waves x 3 ring stages, 4 x 2, 8 x 2
            prev_shape = ME.set_option("hd_shape", shape)
            got.append(run())
            ME.set_option("hd_shape", prev_shape)
    finally:
        ME.set_option("hd_mask", prev)
        ME.set_option("hd_min_rows", prev_rows)
    for k, g in enumerate(got):
        for a, b in zip(want, g):
            assert torch.equal(a, b), "conv_hd (run %d) differs from conv_hl: %d of %d elements, max %g" % (
                k, int((a != b).sum()), a.numel(), float((a - b).abs().max()))
    assert float(want[0].abs().max()) > 0.1


def test_zskip_leaves_every_bit_in_place(cuda, built_lib):
    """Option "zskip" (off by default: measured slower): tiles whose rows have no neighbour in their mask group write no partial
    tile and conv_finish_small reads none for such (group, row) pairs - the sums are the same bits, with and without a
    residual / ReLU / hl epilogue, on conv_hd and conv_hl."""
    coords, _ = scene_coords(5, 40000, small=False)
    N = len(coords)
    rng = np.random.default_rng(11)
    t = lambda a: torch.from_numpy(a).to(cuda)
    cm = ME.CoordinateManager(torch.from_numpy(coords).to(cuda, torch.int32))
    nbr, perms = cm.kernel_map(3, 1), cm.mask_perms(3, 1, 3)
    assert float((nbr.view(N, 3, 9) >= 0).any(2).float().mean()) < 0.95           # some (row, group) pairs ARE empty
    for cin, cout in ((96, 96), (32, 32)):
        x = ME.to_hl(t(rng.normal(0, 1, (N, cin)).astype(np.float32)))
        w = t((rng.normal(0, 1, (27, cin, cout)) / np.sqrt(cin * 27)).astype(np.float32))
        res = ME.to_hl(t(rng.normal(0, 1, (N, cout)).astype(np.float32)))
        shift = t(rng.normal(0, 0.2, cout).astype(np.float32))

        def run():
            plain = ME.conv_forward_masked(x, w, nbr, perms, N, pieces=2, in_hl=True)
            out = torch.empty((N, cout), device=cuda)
            ME.conv_forward_masked(x, w, nbr, perms, N, pieces=2, in_hl=True, shift=shift, residual=res, relu=True, out=out,
                                   out_hl=True, res_hl=True)
            return plain, out

        prev = ME.set_option("zskip", 0)
        try:
            want = run()
            ME.set_option("zskip", 1)
            got = run()
        finally:
            ME.set_option("zskip", prev)
        assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
