"""How many (32-row block, kernel offset) units a fine-level convolution has to multiply under different row orders, and
how large a tile's input window is: the bench scene (80k points), rows in the plan's Z-order.  CPU only.
    python profiles/conv_order_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd.synth import make_scene
sc = make_scene(0, n_points=80000, res=0.03)
c = np.asarray(sc.coords, np.int64)
def spread3(x):
    r = np.zeros_like(x)
    for b in range(6):
        r |= ((x >> b) & 1) << (3 * b)
    return r
def level(c, ts):
    q = (c // ts) * ts
    return np.unique(q, axis=0)
def stats(c, ts, name):
    mn = c.min(0); ex = (c.max(0) - mn).max() // ts
    shift = 0
    while (ex >> shift) >= 64: shift += 1
    g = (c - mn) // ts
    gs = g >> shift
    key = spread3(gs[:, 0]) | (spread3(gs[:, 1]) << 1) | (spread3(gs[:, 2]) << 2)
    order = np.argsort(key, kind='stable')
    c = c[order]; g = g[order]
    n = len(c)
    # neighbour table
    d = {}
    pk = (g[:, 0] << 40) | (g[:, 1] << 20) | g[:, 2]
    idx = dict(zip(pk.tolist(), range(n)))
    offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    nbr = np.full((n, 27), -1, np.int64)
    for j, (dx, dy, dz) in enumerate(offs):
        k = ((g[:, 0] + dx) << 40) | ((g[:, 1] + dy) << 20) | (g[:, 2] + dz)
        nbr[:, j] = [idx.get(v, -1) for v in k.tolist()]
    live = (nbr >= 0)
    print(name, "rows", n, "pairs", live.sum(), "live%% %.1f" % (100 * live.mean()), "shift", shift)
    bits = (live * (1 << np.arange(27))).sum(1)
    for T in (128, 256):
        halo = []; units_ns = []; units_ts = []; units_g3 = []
        for t0 in range(0, n, T):
            rows = slice(t0, min(t0 + T, n))
            nb = nbr[rows]
            u = np.unique(nb[nb >= 0])
            halo.append(len(u) / (rows.stop - rows.start))
            lv = live[rows]
            # no sort: 32-row blocks
            def units(lv):
                tot = 0
                for b in range(0, len(lv), 32):
                    tot += lv[b:b + 32].any(0).sum()
                return tot / (27.0 * ((len(lv) + 31) // 32))
            units_ns.append(units(lv))
            o = np.argsort(bits[rows], kind='stable')
            units_ts.append(units(lv[o]))
            # gray-ish: sort by popcount-major? try sort by first 9 bits then rest (same as lexicographic on reversed)
            hi = (lv[:, ::-1] * (1 << np.arange(27))).sum(1)
            o2 = np.argsort(hi, kind='stable')
            units_g3.append(units(lv[o2]))
        print("  tile %d: halo rows / tile rows mean %.2f max %.2f | units: morton %.3f  tile-local mask sort %.3f  (reverse-bit sort %.3f)" % (T, np.mean(halo), np.max(halo), np.mean(units_ns), np.mean(units_ts), np.mean(units_g3)))
    # global mask sort with 3 groups (current design)
    tot = 0
    for gi in range(3):
        lv = live[:, 9 * gi:9 * gi + 9]
        b9 = (lv * (1 << np.arange(9))).sum(1)
        o = np.argsort(b9, kind='stable')
        l2 = lv[o]
        for b in range(0, n, 32):
            tot += l2[b:b + 32].any(0).sum()
    print("  current: 3 groups, global mask sort: units %.3f" % (tot / (27.0 * ((n + 31) // 32))))
stats(c, 1, "ts1")
stats(level(c, 2), 2, "ts2")

def single_order(c, ts, name):
    mn = c.min(0); g = (c - mn) // ts
    ex = g.max(); shift = 0
    while (ex >> shift) >= 64: shift += 1
    gs = g >> shift
    key = spread3(gs[:, 0]) | (spread3(gs[:, 1]) << 1) | (spread3(gs[:, 2]) << 2)
    order = np.argsort(key, kind='stable'); g = g[order]; n = len(g)
    pk = (g[:, 0] << 40) | (g[:, 1] << 20) | g[:, 2]
    idx = dict(zip(pk.tolist(), range(n)))
    offs = [(dx, dy, dz) for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    live = np.zeros((n, 27), bool)
    for j, (dx, dy, dz) in enumerate(offs):
        k = ((g[:, 0] + dx) << 40) | ((g[:, 1] + dy) << 20) | (g[:, 2] + dz)
        live[:, j] = [v in idx for v in k.tolist()]
    def units(l2):
        tot = 0
        for b in range(0, n, 32):
            tot += l2[b:b + 32].any(0).sum()
        return tot / (27.0 * ((n + 31) // 32))
    bits = (live * (1 << np.arange(27, dtype=np.int64))).sum(1)
    print(name, "single global order by the 27-bit mask: units %.3f" % units(live[np.argsort(bits, kind='stable')]))
    # order by popcount-weighted: sort by mask of the 13 "most common" offsets first
    freq = live.mean(0); rank = np.argsort(-freq)
    b2 = (live[:, rank[::-1]] * (1 << np.arange(27, dtype=np.int64))).sum(1)
    print(name, "  ... most frequent offsets as the most significant bits: units %.3f" % units(live[np.argsort(b2, kind='stable')]))
    b3 = (live[:, rank] * (1 << np.arange(27, dtype=np.int64))).sum(1)
    print(name, "  ... least frequent offsets as the most significant bits: units %.3f" % units(live[np.argsort(b3, kind='stable')]))
single_order(c, 1, "ts1")
single_order(level(c, 2), 2, "ts2")
