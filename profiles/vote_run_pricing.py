"""VERDICT r5 item 3a, priced on the CPU before building anything: how long are the RUNS of consecutive votes of one point
(consecutive rotations - what a lane of hv_fwd_tiles walks, csrc/hv_vote.hip "lane l walks items [l*S, (l+1)*S)") that keep the
same floor cell (x, z)?  A run could be merged in registers (integer adds of the 2^-36 fixed-point contributions: bit-identical)
and cost one set of 24 LDS atomics instead of one per vote.  Also: the bank-pair load of 64 random votes of a tile (what a
drain64 instruction sees), the bound a perfect in-wave reordering could reach.

    python profiles/vote_run_pricing.py [seed] [n_points]        (bench.py's scene 0: seed 0, 80 000 points, teacher predictions)
"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from canonicalvoting_amd.synth import make_scene, synth_predictions  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
R, res = 120, np.float32(0.03)
sc = make_scene(seed, n_points=n)
xyz, scale, prob, cls = synth_predictions(sc)
pts = sc.points.astype(np.float32)
corner = pts.min(0)
dims = (np.trunc((pts.max(0) - corner) / res) + 1).astype(np.int64)
corr = (xyz * scale).astype(np.float32)
step = np.float32(2 * 3.141592654 / R)
theta = (np.arange(R, dtype=np.float32) * step).astype(np.float32)
c, s = np.cos(theta).astype(np.float32), np.sin(theta).astype(np.float32)
ox = (-c[None] * corr[:, 0:1] + s[None] * corr[:, 2:3]).astype(np.float32)
oz = (-s[None] * corr[:, 0:1] - c[None] * corr[:, 2:3]).astype(np.float32)
gx = ((pts[:, 0:1] + ox - corner[0]) / res).astype(np.float32)
gz = ((pts[:, 2:3] + oz - corner[2]) / res).astype(np.float32)
gy = ((pts[:, 1] - corr[:, 1] - corner[1]) / res).astype(np.float32)
inb = (gx >= 0) & (gz >= 0) & (gx < dims[0] - 1) & (gz < dims[2] - 1) & ((gy >= 0) & (gy < dims[1] - 1))[:, None]
fx, fz = gx.astype(np.int64), gz.astype(np.int64)
print("scene seed %d, %d points, grid %s, in-bounds votes %d (%.1f%%)" % (seed, n, tuple(dims), inb.sum(), 100 * inb.mean()))
# runs along the rotation axis (cyclic): vote r continues the run of vote r-1 when both are in bounds and in the same floor cell
same = inb & np.roll(inb, 1, axis=1) & (fx == np.roll(fx, 1, axis=1)) & (fz == np.roll(fz, 1, axis=1))
votes = inb.sum()
runs = votes - same.sum()
print("same floor cell as the previous rotation: %.1f%% of the votes -> mean run length %.3f (runs %d)" % (100 * same.sum() / votes, votes / runs, runs))
# a lane's segment is also cut by the TILE (16 x 32 cells) and the chunk's S items: both only shorten runs
TX, TZ = 16, 32
same_t = same & (fx // TX == np.roll(fx, 1, axis=1) // TX) & (fz // TZ == np.roll(fz, 1, axis=1) // TZ)
print("(identical with the tile cut: %.3f - a run never crosses a tile edge)" % (votes / (votes - same_t.sum())))
rad = np.sqrt(corr[:, 0] ** 2 + corr[:, 2] ** 2) / res
w = inb.sum(1)
print("ring radius in cells (vote-weighted): mean %.1f, median %.1f; arc step per rotation = radius x %.4f -> %.2f cells at the median"
      % (np.average(rad, weights=w), np.median(np.repeat(rad, w)), step, np.median(np.repeat(rad, w)) * step))
for lo, hi in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 1e9)):
    m = (rad >= lo) & (rad < hi)
    v = inb[m].sum()
    if v:
        print("  radius %4g-%-4g cells: %5.1f%% of the votes, run length %.2f" % (lo, hi, 100 * v / votes, v / max(1, v - same[m].sum())))
# what merging would save: LDS atomics per vote stay 24 per RUN; the conversions (24 per vote) stay
print("LDS atomics with run merging: %.1f%% of today's" % (100 * runs / votes))

# ---- bank-pair load of a drain64: 64 votes of one (tile, plane) workgroup, corner (0, 0) word of a channel
# word = cx * 40 + cz (ACC_PITCH 40), a 64-bit word sits on bank pair (word mod 32)
rng = np.random.default_rng(0)
yb = gy.astype(np.int64)
sel = np.nonzero(inb)
order = rng.permutation(len(sel[0]))[:2_000_000]
px, pr = sel[0][order], sel[1][order]
key = (yb[px] * 64 + fx[px, pr] // TX) * 64 + fz[px, pr] // TZ           # (plane, tile)
lx, lz = fx[px, pr] % TX, fz[px, pr] % TZ
o = np.argsort(key, kind="stable")
key, lx, lz = key[o], lx[o], lz[o]
bounds = np.flatnonzero(np.diff(key)) + 1
starts = np.concatenate([[0], bounds])
ends = np.concatenate([bounds, [len(key)]])
cyc_now, cyc_best, cyc_ideal, groups = 0.0, 0.0, 0.0, 0
for a, b in zip(starts, ends):
    for g0 in range(a, b - 63, 64):
        bank = (lx[g0:g0 + 64] * 40 + lz[g0:g0 + 64]) % 32
        # today: lanes in queue order, two halves of 32 lanes; a half takes max-load cycles (distinct addresses on one bank pair
        # serialise; same-address lanes are counted as serialised too - an upper bound on what reordering can win)
        h = [np.bincount(bank[k:k + 32], minlength=32).max() for k in (0, 32)]
        cyc_now += h[0] + h[1]
        cyc_best += max(2, np.bincount(bank, minlength=32).max())      # any split of the 64 into two halves needs >= the max load
        cyc_ideal += 2
        groups += 1
        if groups >= 20000:
            break
    if groups >= 20000:
        break
print("drain64 bank-pair load over %d groups of 64 votes of one (tile, plane): queue order %.2f cycles per instruction, "
      "perfect reordering inside the 64 >= %.2f, conflict-free 2.00" % (groups, cyc_now / groups, cyc_best / groups))
print("-> a perfect bank-aware permutation of a drain's 64 votes can save at most %.0f%% of the atomic cycles" % (100 * (1 - cyc_best / cyc_now)))
