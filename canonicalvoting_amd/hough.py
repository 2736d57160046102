"""``HVFunction`` / ``HoughVoting`` exactly as the four reference scripts define them
(eval_joint.py:24-57, train_joint.py:22-56, train_separate.py:23-56, eval_separate.py:19-44;
7-argument variant sunrgbd/brnetcanon.py:94-117) - one importable copy instead of four."""
import torch

from . import hv_cuda


class HVFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, xyz, scale, obj, res, num_rots, corners=None):
        ctx.save_for_backward(points, xyz, scale, obj, res, num_rots)
        # the 7-argument variant votes into a grid anchored at corners[0]; its backward has to sample grad_grid
        # at the same cells (the reference's 7-argument Function, sunrgbd/brnetcanon.py:94-103, defines no
        # backward at all: it raises instead of returning numbers)
        ctx.corners = None if corners is None else corners.detach()
        if corners is None:
            outputs = hv_cuda.forward(points, xyz, scale, obj, res, num_rots)
        else:
            outputs = hv_cuda.forward(points, xyz, scale, obj, res, num_rots, corners)
        grid_obj, grid_rot, grid_scale = outputs
        return grid_obj, grid_rot, grid_scale

    @staticmethod
    def backward(ctx, grad_obj, grad_rot, grad_scale):
        # only grad_obj is propagated (eval_joint.py:33-38); grad_rot / grad_scale are ignored
        points, xyz, scale, obj, res, num_rots = ctx.saved_tensors
        outputs = hv_cuda.backward(grad_obj.contiguous(), points, xyz, scale, obj, res, num_rots,
                                   corners=ctx.corners)
        d_xyz_labels, d_scale_labels, d_obj_labels = outputs
        return None, d_xyz_labels, d_scale_labels, d_obj_labels, None, None, None


class HoughVoting(torch.nn.Module):
    def __init__(self, res=0.03, num_rots=120):
        super().__init__()
        self.res = torch.tensor(res, dtype=torch.float32).cuda()
        self.num_rots = torch.tensor(num_rots, dtype=torch.int32).cuda()

    def forward(self, points, xyz, scale, obj, corners=None):
        if corners is None:
            return HVFunction.apply(points, xyz, scale, obj, self.res, self.num_rots)
        return HVFunction.apply(points, xyz, scale, obj, self.res, self.num_rots, corners)
