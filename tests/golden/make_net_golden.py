"""Golden vectors for the network composition, produced by executing the reference's own class definitions
(utils/resnet.py:67-154 ResNetBase / _make_layer, utils/minkunet.py:36-180 MinkUNetBase.network_initialization and
forward, :244 MinkUNet34C) in the build container.  MinkowskiEngine v0.5.3 is absent from the image, so the names the
reference imports from it (MinkowskiConvolution, MinkowskiConvolutionTranspose, MinkowskiBatchNorm, MinkowskiReLU,
cat, modules.resnet_block.BasicBlock ...) are bound here to thin nn.Modules whose arithmetic is the oracle's primitive
ops (oracle/sparse_oracle.py conv / conv_transpose_k2s2 / kernel_map, themselves pinned against dense F.conv3d in
tests/test_sparse_oracle.py).  What this pins is everything the reference's Python decides: layer order, channel
widths, strides, which tensors are concatenated and added, BN/ReLU placement, parameter names and shapes (the state
dict is loaded with strict=True into the reference's module tree).  It does NOT pin the primitive arithmetic of
MinkowskiEngine itself - that stays "parity unpinned" (oracle/sparse_oracle.py header).

Stores (tests/golden/net_ref.npz): the eval-mode and training-mode outputs of reference MinkUNet34C(3, 64).forward on
a seeded 900-point scene with oracle.make_state_dict(seed=11) weights, the state-dict names/shapes of the reference
module tree, and the sequence of primitive calls its forward made.

    python tests/golden/make_net_golden.py            # needs /root/reference
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

SEED_W, SEED_SCENE, N = 11, 21, 900


def make_inputs():
    from canonicalvoting_amd.synth import make_scene
    sc = make_scene(SEED_SCENE, n_points=N, res=0.06, room=(1.8, 1.0, 1.6), n_boxes=2, margin=0.5, box_scale=0.4)
    coords = np.concatenate([np.zeros((N, 1), np.int64), sc.coords], 1)
    return coords, (sc.feats * 2 - 1).astype(np.float32)


def engine_standin(trace):
    """A module object answering to the MinkowskiEngine names utils/resnet.py and utils/minkunet.py use."""
    from oracle import sparse_oracle as so
    ME = types.ModuleType("MinkowskiEngine")

    class SparseTensor:
        def __init__(self, F, cm, ts):
            self.F, self.cm, self.ts = F, cm, ts

        def __add__(self, o):                           # `out += residual` of BasicBlock
            trace.append("add")
            return SparseTensor(self.F + o.F, self.cm, self.ts)
        __iadd__ = __add__

    class MinkowskiConvolution(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
            super().__init__()
            assert dimension == 3 and dilation == 1
            K = kernel_size ** 3
            self.k, self.stride = kernel_size, stride
            self.kernel = nn.Parameter(torch.zeros((in_channels, out_channels) if K == 1 else (K, in_channels, out_channels)))
            self.bias = nn.Parameter(torch.zeros(1, out_channels)) if bias else None

        def forward(self, x):
            trace.append("conv k%d s%d %d->%d @%d" % (self.k, self.stride, self.kernel.shape[-2], self.kernel.shape[-1], x.ts))
            y = so.conv(x.F, self.kernel.detach(), x.cm.map(self.k, x.ts, self.stride),
                        None if self.bias is None else self.bias.detach())
            return SparseTensor(y, x.cm, x.ts * self.stride)

    class MinkowskiConvolutionTranspose(MinkowskiConvolution):
        def forward(self, x):
            assert self.k == 2 and self.stride == 2
            trace.append("convtr k2 s2 %d->%d @%d" % (self.kernel.shape[-2], self.kernel.shape[-1], x.ts))
            y = so.conv_transpose_k2s2(x.F, self.kernel.detach(), x.cm.map(2, x.ts // 2, 2))
            return SparseTensor(y, x.cm, x.ts // 2)

    class MinkowskiBatchNorm(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1):
            super().__init__()
            self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

        def forward(self, x):
            trace.append("bn %d" % self.bn.num_features)
            return SparseTensor(self.bn(x.F), x.cm, x.ts)

    class MinkowskiReLU(nn.Module):
        def __init__(self, inplace=False):
            super().__init__()

        def forward(self, x):
            trace.append("relu")
            return SparseTensor(torch.relu(x.F), x.cm, x.ts)

    class _Unused(nn.Module):                           # ResNetBase constructs none of these for MinkUNet
        def __init__(self, *a, **k):
            super().__init__()

    def cat(*xs):
        trace.append("cat " + "+".join(str(x.F.shape[1]) for x in xs))
        return SparseTensor(torch.cat([x.F for x in xs], 1), xs[0].cm, xs[0].ts)

    class BasicBlock(nn.Module):
        """MinkowskiEngine v0.5.3 MinkowskiEngine/modules/resnet_block.py BasicBlock as published:
        conv3(stride)-norm1-relu-conv3-norm2, optional downsample of the input, add, relu."""
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
            super().__init__()
            self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
            self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
            self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, dimension=dimension)
            self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
            self.relu = MinkowskiReLU(inplace=True)
            self.downsample = downsample

        def forward(self, x):
            residual = x
            out = self.relu(self.norm1(self.conv1(x)))
            out = self.norm2(self.conv2(out))
            if self.downsample is not None:
                residual = self.downsample(x)
            out += residual
            return self.relu(out)

    ME.SparseTensor = SparseTensor
    ME.MinkowskiConvolution = MinkowskiConvolution
    ME.MinkowskiConvolutionTranspose = MinkowskiConvolutionTranspose
    ME.MinkowskiBatchNorm = MinkowskiBatchNorm
    ME.MinkowskiReLU = MinkowskiReLU
    ME.MinkowskiAvgPooling = ME.MinkowskiGlobalMaxPooling = ME.MinkowskiLinear = _Unused
    ME.cat = cat
    ME.utils = types.SimpleNamespace(kaiming_normal_=lambda t, **k: t, batched_coordinates=None)
    mods = types.ModuleType("MinkowskiEngine.modules")
    rb = types.ModuleType("MinkowskiEngine.modules.resnet_block")
    rb.BasicBlock, rb.Bottleneck = BasicBlock, _Unused
    mods.resnet_block = rb
    ME.modules = mods
    return {"MinkowskiEngine": ME, "MinkowskiEngine.modules": mods, "MinkowskiEngine.modules.resnet_block": rb}


if __name__ == "__main__":
    assert os.path.isdir("/root/reference/utils"), "run where /root/reference is mounted"
    from oracle import sparse_oracle as so
    trace = []
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.resnet", "utils.minkunet")}
    sys.modules.update(engine_standin(trace))
    sys.path.insert(0, "/root/reference")
    sys.dont_write_bytecode = True
    try:
        from utils.minkunet import MinkUNet34C                         # the reference's class, executed as it lies
        ME = sys.modules["MinkowskiEngine"]
        torch.manual_seed(0)
        net = MinkUNet34C(3, 64)
        sd = so.make_state_dict(3, 64, seed=SEED_W)
        net.load_state_dict(sd, strict=True)                           # names and shapes must agree exactly
        names = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        coords, feats = make_inputs()
        outs = {}
        for mode in ("eval", "train"):
            net.train(mode == "train")
            del trace[:]
            with torch.no_grad():
                y = net(ME.SparseTensor(torch.from_numpy(feats), so.CoordinateManager(coords), 1))
            outs[mode] = y.F.numpy().astype(np.float32)
            outs[mode + "_trace"] = list(trace)
        assert outs["eval_trace"] == outs["train_trace"]
        np.savez_compressed(os.path.join(HERE, "net_ref.npz"), seed_w=SEED_W, seed_scene=SEED_SCENE, n=N,
                            out_eval=outs["eval"], out_train=outs["train"],
                            state_dict=json.dumps(names), trace=json.dumps(outs["eval_trace"]))
        print("ops", len(trace), "params", len(names), "out", outs["eval"].shape, float(np.abs(outs["eval"]).mean()))
    finally:
        sys.path.remove("/root/reference")
        for k in ("utils", "utils.resnet", "utils.minkunet"):
            sys.modules.pop(k, None)
            if saved[k] is not None:
                sys.modules[k] = saved[k]
