"""MinkUNet family with the reference's module tree, so state-dict keys match
(topology utils/minkunet.py:50-120, forward :122-180; residual layers
utils/resnet.py:118-154; init utils/resnet.py:109-116).  The scripts only use
``MinkUNet34C(in_channels, out_channels, D=3)`` (train_joint.py:218, eval_joint.py:151).

Two forwards with identical results:
  * module-by-module through the MinkowskiEngine facade (works in train and eval mode,
    one launch per conv / batch-norm / relu, like the reference);
  * ``fused_forward`` (taken automatically in eval mode under ``torch.no_grad()``): every
    conv runs with the eval-mode BatchNorm affine, bias, residual add and ReLU folded into
    its epilogue, skip connections are written straight into the concat buffers
    (no ``cat`` copy): 63 launches for the whole network.
"""
import os

import torch
import torch.nn as nn

from . import me as ME
from .me.modules.resnet_block import BasicBlock


class ResNetBase(nn.Module):
    BLOCK = None
    LAYERS = ()
    INIT_DIM = 64
    PLANES = (64, 128, 256, 512)

    def __init__(self, in_channels, out_channels, D=3):
        nn.Module.__init__(self)
        self.D = D
        assert self.BLOCK is not None
        self.network_initialization(in_channels, out_channels, D)
        self.weight_initialization()

    def weight_initialization(self):
        # utils/resnet.py:109-116: kaiming-normal (fan_out, relu) on every MinkowskiConvolution
        # kernel (transposed convs are a different class and keep their default), BN gamma 1 beta 0
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, bn_momentum=0.1):
        # utils/resnet.py:118-154
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                ME.MinkowskiConvolution(self.inplanes, planes * block.expansion, kernel_size=1,
                                        stride=stride, dimension=self.D),
                ME.MinkowskiBatchNorm(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=downsample,
                        dimension=self.D)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, stride=1, dilation=dilation, dimension=self.D)
                   for _ in range(1, blocks)]
        return nn.Sequential(*layers)


_DOWN = (("conv1p1s2", "bn1"), ("conv2p2s2", "bn2"), ("conv3p4s2", "bn3"), ("conv4p8s2", "bn4"))
_UP = (("convtr4p16s2", "bntr4"), ("convtr5p8s2", "bntr5"), ("convtr6p4s2", "bntr6"),
       ("convtr7p2s2", "bntr7"))


class MinkUNetBase(ResNetBase):
    BLOCK = None
    PLANES = None
    DILATIONS = (1,) * 8
    LAYERS = (2,) * 8
    INIT_DIM = 32
    OUT_TENSOR_STRIDE = 1

    def network_initialization(self, in_channels, out_channels, D):
        exp = self.BLOCK.expansion
        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes)
        for i, (cname, bname) in enumerate(_DOWN):              # encoder: ts 1 -> 16
            setattr(self, cname, ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2,
                                                         stride=2, dimension=D))
            setattr(self, bname, ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, "block%d" % (i + 1), self._make_layer(self.BLOCK, self.PLANES[i], self.LAYERS[i]))
        skip = (self.PLANES[2] * exp, self.PLANES[1] * exp, self.PLANES[0] * exp, self.INIT_DIM)
        for i, (cname, bname) in enumerate(_UP):                # decoder: ts 16 -> 1
            setattr(self, cname, ME.MinkowskiConvolutionTranspose(self.inplanes, self.PLANES[4 + i],
                                                                  kernel_size=2, stride=2, dimension=D))
            setattr(self, bname, ME.MinkowskiBatchNorm(self.PLANES[4 + i]))
            self.inplanes = self.PLANES[4 + i] + skip[i]
            setattr(self, "block%d" % (5 + i),
                    self._make_layer(self.BLOCK, self.PLANES[4 + i], self.LAYERS[4 + i]))
        self.final = ME.MinkowskiConvolution(self.PLANES[7] * exp, out_channels, kernel_size=1, bias=True,
                                             dimension=D)
        self.relu = ME.MinkowskiReLU(inplace=True)

    # ------------------------------------------------------------------ reference-shaped forward
    def forward(self, x):
        if (not self.training) and (not torch.is_grad_enabled()) and self.BLOCK is BasicBlock:
            return self.fused_forward(x)
        return self.modular_forward(x)

    def modular_forward(self, x):
        out_p1 = self.bn0.forward_fused(self.conv0p1s1(x), relu=True)
        skips = [out_p1]
        out = out_p1
        for i, (cname, bname) in enumerate(_DOWN):
            out = getattr(self, bname).forward_fused(getattr(self, cname)(out), relu=True)
            out = getattr(self, "block%d" % (i + 1))(out)
            skips.append(out)
        skips.pop()                                             # block4 output is not a skip
        for i, (cname, bname) in enumerate(_UP):
            out = getattr(self, bname).forward_fused(getattr(self, cname)(out), relu=True)
            out = ME.cat(out, skips.pop())
            out = getattr(self, "block%d" % (5 + i))(out)
        return self.final(out)

    # ------------------------------------------------------------------ fused eval forward
    def _fold(self, bn_module):
        bn = bn_module.bn
        ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               bn.weight.data_ptr())
        cache = self.__dict__.setdefault("_fold_cache", {})
        hit = cache.get(id(bn))
        if hit is None or hit[0] != ver:
            hit = (ver, ME.bn_affine(bn))
            cache[id(bn)] = hit
        return hit[1]

    # levels with at least this many rows run the mask-sorted grouped conv
    MASKED_MIN_ROWS = int(os.environ.get("CV_MASKED_MIN_ROWS", "16384"))
    MASK_GROUPS = 4

    def _conv3(self, x, kernel, nbr, perms, n, **ep):
        if perms is not None:
            return ME.conv_forward_masked(x, kernel, nbr, perms, n, **ep)
        return ME.conv_forward(x, kernel, nbr, n, **ep)

    def _run_layer(self, layer, x, nbr, n, out_view, perms=None):
        """Sequential of BasicBlocks on features x ([n, C] view); the last block writes out_view."""
        for bi, blk in enumerate(layer):
            s1, b1 = self._fold(blk.norm1)
            t = self._conv3(x, blk.conv1.kernel, nbr, perms, n, scale=s1, shift=b1, relu=True)
            if blk.downsample is not None:
                sd, bd = self._fold(blk.downsample[1])
                res = ME.conv_forward(x, blk.downsample[0].kernel, None, n, scale=sd, shift=bd)
            else:
                res = x
            s2, b2 = self._fold(blk.norm2)
            last = bi == len(layer) - 1
            x = self._conv3(t, blk.conv2.kernel, nbr, perms, n, scale=s2, shift=b2, residual=res, relu=True,
                            out=out_view if (last and out_view is not None) else None)
        return x

    def fused_forward(self, x):
        # internal rows are Z-order sorted (compact 32-row wave tiles -> whole kernel offsets are
        # skipped); the stem reads the caller's rows through stem_map and `final` writes back in the
        # caller's row order through out_map, so the row-order invariant of the reference holds.
        cm, stem_map, out_map = x.coordinate_manager.fused_plan()
        dev = x.F.device
        exp = self.BLOCK.expansion
        n = [cm.num_rows(1 << i) for i in range(5)]
        perms = lambda ts, rows: cm.mask_perms(3, ts, self.MASK_GROUPS) if rows >= self.MASKED_MIN_ROWS else None
        # concat buffers of the decoder: [convtr output | encoder skip]
        up_c = [self.PLANES[4 + i] for i in range(4)]
        skip_c = (self.PLANES[2] * exp, self.PLANES[1] * exp, self.PLANES[0] * exp, self.INIT_DIM)
        cat = [torch.empty((n[3 - i], up_c[i] + skip_c[i]), dtype=torch.float32, device=dev) for i in range(4)]
        skip_view = [cat[i][:, up_c[i]:] for i in range(4)]     # i=3 <- out_p1, 2 <- block1, 1 <- block2, 0 <- block3
        s, b = self._fold(self.bn0)
        out = ME.conv_forward(x.F.contiguous(), self.conv0p1s1.kernel, stem_map, n[0], scale=s,
                              shift=b, relu=True, out=skip_view[3])
        for i, (cname, bname) in enumerate(_DOWN):
            ts = 1 << i
            s, b = self._fold(getattr(self, bname))
            out = ME.conv_forward(out, getattr(self, cname).kernel, cm.kernel_map(2, ts, 2), n[i + 1],
                                  scale=s, shift=b, relu=True)
            out = self._run_layer(getattr(self, "block%d" % (i + 1)), out, cm.kernel_map(3, 2 * ts), n[i + 1],
                                  skip_view[2 - i] if i < 3 else None, perms(2 * ts, n[i + 1]))
        for i, (cname, bname) in enumerate(_UP):
            ts_coarse = 16 >> i
            lvl = 3 - i
            s, b = self._fold(getattr(self, bname))
            ME.conv_forward(out, getattr(self, cname).kernel, cm.up_map(ts_coarse), n[lvl], scale=s, shift=b,
                            relu=True, out=cat[i][:, :up_c[i]], row_perm=cm.up_perm(ts_coarse))
            out = self._run_layer(getattr(self, "block%d" % (5 + i)), cat[i], cm.kernel_map(3, ts_coarse // 2),
                                  n[lvl], None, perms(ts_coarse // 2, n[lvl]))
        y = ME.conv_forward(out, self.final.kernel, out_map, n[0], shift=self.final.bias.reshape(-1))
        return x._like(y, 1)


    def forward_flops(self, x):
        """Algorithmic flops of one forward on x's coordinate set:
        sum over conv layers of 2 * P * Cin * Cout with P = existing (input, output) pairs of the
        layer's kernel map, counted exactly from the maps.  Also returns the dense-equivalent count
        (every kernel offset present)."""
        cm = x.coordinate_manager.fused_plan()[0]
        n = [cm.num_rows(1 << i) for i in range(5)]
        pairs = lambda m: int((m >= 0).sum().item())
        p5, p3 = pairs(x.coordinate_manager.fused_plan()[1]), {1 << i: pairs(cm.kernel_map(3, 1 << i)) for i in range(5)}
        pd = {1 << i: pairs(cm.kernel_map(2, 1 << i, 2)) for i in range(4)}
        sparse = dense = 0

        def add(p, full, cin, cout):
            nonlocal sparse, dense
            sparse += 2 * p * cin * cout
            dense += 2 * full * cin * cout

        def layer(seq, ts, lvl):
            for blk in seq:
                cin, cout = blk.conv1.in_channels, blk.conv1.out_channels
                add(p3[ts], n[lvl] * 27, cin, cout)
                add(p3[ts], n[lvl] * 27, cout, cout)
                if blk.downsample is not None:
                    add(n[lvl], n[lvl], cin, cout)

        add(p5, n[0] * 125, self.conv0p1s1.in_channels, self.conv0p1s1.out_channels)
        for i, (cname, _) in enumerate(_DOWN):
            c = getattr(self, cname)
            add(pd[1 << i], n[i + 1] * 8, c.in_channels, c.out_channels)
            layer(getattr(self, "block%d" % (i + 1)), 2 << i, i + 1)
        for i, (cname, _) in enumerate(_UP):
            c = getattr(self, cname)
            lvl = 3 - i
            add(n[lvl], n[lvl], c.in_channels, c.out_channels)          # one parent per fine voxel
            layer(getattr(self, "block%d" % (5 + i)), 8 >> i, lvl)
        add(n[0], n[0], self.final.in_channels, self.final.out_channels)
        return sparse, dense


class MinkUNet14(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class MinkUNet18(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)


class MinkUNet34(MinkUNetBase):
    BLOCK = BasicBlock
    LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class MinkUNet14A(MinkUNet14):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet18A(MinkUNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet18B(MinkUNet18):
    PLANES = (32, 64, 128, 256, 128, 128, 128, 128)


class MinkUNet18D(MinkUNet18):
    PLANES = (32, 64, 128, 256, 384, 384, 384, 384)


class MinkUNet34A(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 64)


class MinkUNet34B(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 64, 32)


class MinkUNet34C(MinkUNet34):
    PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


# ----------------------------------------------------------------------------------------------- checkpoints
def convert_kernel_offset_order(state_dict, from_order="z_fastest", to_order="x_fastest"):
    """Checkpoint interchange.  Parameter names and shapes are MinkowskiEngine's, so the
    authors' ``joint.pth`` / ``separate/*.pth`` load with ``load_state_dict`` as they are.  The one thing that
    cannot be verified without a real checkpoint is the order in which ME enumerates the K^3 kernel offsets
    along dim 0 of every ``kernel``; this engine runs the first spatial axis fastest (``x_fastest``).  If ME's
    order turns out to be the last axis fastest, this permutes every [K^3, Cin, Cout] kernel accordingly
    (k = round(K^(1/3)); 1x1 kernels are untouched).  Returns a new dict."""
    if from_order == to_order:
        return dict(state_dict)
    out = {}
    for name, w in state_dict.items():
        if name.endswith(".kernel") and w.dim() == 3:
            K = w.shape[0]
            k = int(round(K ** (1.0 / 3.0)))
            assert k ** 3 == K, name
            # index j = a + k*(b + k*c) with (a,b,c) = (x,y,z) for x_fastest and (z,y,x) for z_fastest
            w = w.reshape(k, k, k, *w.shape[1:]).permute(2, 1, 0, 3, 4).reshape(K, *w.shape[1:]).contiguous()
        out[name] = w
    return out


def load_reference_checkpoint(model, path, offset_order="x_fastest", key=None):
    """torch.load + (optional) kernel-offset permutation + load_state_dict.  ``key`` selects a sub-dict such
    as ``model_state_dict`` of the SUN RGB-D checkpoint (sunrgbd/brnetcanon.py:167)."""
    sd = torch.load(path, map_location="cpu")
    if key is not None:
        sd = sd[key]
    model.load_state_dict(convert_kernel_offset_order(sd, offset_order, "x_fastest"))
    return model
