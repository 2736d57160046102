"""SUN RGB-D proposal sampler of the reference's BRNet plug-in:
``HoughVotingModule`` of sunrgbd/brnetcanon.py:104-162 with the same constructor, forward signature and
return values.  The vote is the 7-argument ``hv_cuda.forward(..., corners)`` (brnetcanon.py:99); everything
after it (max / argmax over the up axis, power, multinomial sampling of 1.5 x num_proposal cells, rejection
by distance to the seed votes, truncation) stays on the device as torch ops - it runs once per scene on a
[X, Z] map and is not a hot loop.  mmdet3d / BRNet themselves are out of scope (absent dependencies)."""
import torch
import torch.nn as nn

from .hough import HVFunction


def unravel_index(index, shape):
    """sunrgbd/brnetcanon.py:86-91"""
    out = []
    for dim in reversed(shape):
        out.append(index % dim)
        index = torch.div(index, dim, rounding_mode="floor")
    return tuple(reversed(out))


class HoughVotingModule(nn.Module):
    def __init__(self, res=0.03, num_rots=36, nms_size=0.15, thresh=0, num_proposal=256, no_grad=True):
        super().__init__()
        self.res = torch.tensor(res, dtype=torch.float32, device="cuda")
        self.num_rots = torch.tensor(num_rots, dtype=torch.int32, device="cuda")
        self.no_grad = no_grad
        self.nms_size_grid = int(nms_size // res)
        self.num_proposal = num_proposal
        self.thresh = thresh

    def _sample(self, dist, n):
        """the one stochastic step (brnetcanon.py:137); a method so tests can pin the draws"""
        return torch.multinomial(dist, n, replacement=True)

    def forward(self, pc, xyz, scale, prob, corners, vote_points, pow=0.5):
        with torch.set_grad_enabled(not self.no_grad):
            hv_map, _, hv_scale = HVFunction.apply(pc.contiguous(), xyz.contiguous(), scale.contiguous(),
                                                   prob.contiguous(), self.res, self.num_rots, corners)
        hv_map_y = hv_map.max(1)[0] + 1e-7                                   # :124
        hv_map_y = torch.pow(hv_map_y, pow)
        hv_map_yidx = torch.argmax(hv_map, 1)
        dist = hv_map_y.reshape(-1)
        if (not torch.all(torch.isfinite(dist))) or (dist.sum() < 1e-7):     # :129-130
            dist = torch.ones_like(dist)
        cnt = 0
        loc, sample_vals, scales = [], [], []
        while cnt < self.num_proposal:                                       # :135-153
            sample = self._sample(dist, int(self.num_proposal * 1.5))
            sample_val = dist[sample]
            ix, iz = unravel_index(sample, hv_map_y.shape)
            iy = hv_map_yidx[ix, iz]
            world_loc = torch.stack([ix, iy, iz], -1) * self.res + corners[0]
            sc = hv_scale[ix, iy, iz, :]
            dist2seed = torch.min(torch.cdist(world_loc, vote_points), -1)[0]
            near = dist2seed < 0.3
            if torch.sum(near) == 0:
                loc.append(world_loc); sample_vals.append(sample_val); scales.append(sc)
            else:
                loc.append(world_loc[near]); sample_vals.append(sample_val[near]); scales.append(sc[near])
            cnt += loc[-1].shape[0]
        candidates = torch.cat(loc)[:self.num_proposal]
        scales = torch.cat(scales)[:self.num_proposal]
        probs = torch.zeros_like(candidates)[..., 0]                         # :160 (the reference returns zeros)
        return candidates, probs, scales
