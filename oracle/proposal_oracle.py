"""CPU restatement of the reference's SUN RGB-D proposal sampler - TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Follows sunrgbd/brnetcanon.py:114-162 line by line in numpy, given the vote grids and the sequence of multinomial
draws (the only stochastic step, :137).  PINNED BY REFERENCE EXECUTION: the module cannot be imported here (mmdet3d /
BRNet / cv2 / visdom are absent), so tests/golden/make_proposal_golden.py exec()s the class's own lines on CPU torch
with the draws recorded; tests/test_oracle_vote.py checks this file against the result."""
import numpy as np


def sample_proposals(hv_map, hv_scale, corner0, res, vote_points, draws, num_proposal, pow=0.5):
    """hv_map [X,Y,Z], hv_scale [X,Y,Z,3] float32; draws: list of int arrays (one per loop trip)."""
    hv_map = np.asarray(hv_map, np.float32)
    hv_map_y = (hv_map.max(1) + np.float32(1e-7)).astype(np.float32)           # :124
    hv_map_y = np.power(hv_map_y, np.float32(pow)).astype(np.float32)
    yidx = hv_map.argmax(1)
    dist = hv_map_y.reshape(-1)
    uniform = (not np.all(np.isfinite(dist))) or dist.sum() < 1e-7
    if uniform:
        dist = np.ones_like(dist)
    Z = hv_map_y.shape[1]
    loc, scales, cnt, used = [], [], 0, 0
    while cnt < num_proposal:
        s = np.asarray(draws[used], np.int64); used += 1
        ix, iz = s // Z, s % Z
        iy = yidx[ix, iz]
        world = (np.stack([ix, iy, iz], -1).astype(np.float32) * np.float32(res) + np.asarray(corner0, np.float32))
        sc = hv_scale[ix, iy, iz, :]
        d = np.sqrt(((world[:, None, :].astype(np.float64) - np.asarray(vote_points, np.float64)[None]) ** 2).sum(-1)).min(-1)
        near = d < 0.3
        if near.sum() == 0:
            loc.append(world); scales.append(sc)
        else:
            loc.append(world[near]); scales.append(sc[near])
        cnt += len(loc[-1])
    return np.concatenate(loc)[:num_proposal], np.concatenate(scales)[:num_proposal], dist, used
