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

    def train(self, mode=True):
        # every derived cache (folded BatchNorm affines, packed weights, the C program) is keyed by tensor versions,
        # which fused optimizers do not bump: drop them whenever the mode is switched
        self.__dict__.pop("_fold_cache", None)
        self.__dict__.pop("_prog", None)
        ME.invalidate_weight_caches()
        return super().train(mode)

    # ------------------------------------------------------------------ reference-shaped forward
    def forward(self, x, defer_check=None):
        if (not self.training) and (not torch.is_grad_enabled()) and self.BLOCK is BasicBlock:
            return self.program_forward(x, defer_check=defer_check) if self.USE_PROGRAM else self.fused_forward(x)
        with ME.batched_counter_updates():
            return self.modular_forward(x)

    # Training / autograd forward on the SPATIALLY SORTED twin of the coordinate set (round 4): the caller's rows arrive in
    # arbitrary order (sparse_quantize keeps the point cloud's; synthetic scenes are random), so every gather of the 63
    # forward, 63 input-gradient and 63 weight-gradient launches of a step was a random one.  As the eval paths do, the stem
    # reads the caller's rows through its map and writes Z-order rows, `final` writes back through the inverse order (the
    # row-order invariant of train_joint.py:256-272 holds), and every level, kernel map and mask order comes from ONE
    # cv_sp_scene_plan call instead of ~25 lazy ones.  Measured (profiles/r4/train_sorted_ab.txt, random row order in): 31.9-32.0 ->
    # 31.2-31.4 ms per step on conv_rows_wp; with the hl-format kernels underneath (round 5) 27.4-27.9 -> 26.5-27.2 ms
    # (profiles/r5/train_sorted.txt): the default since then (CV_TRAIN_SORTED=0 switches it off; the gradient tests that hand ReLU
    # masks to the oracle set SORTED_TRAINING = False on their model: they read the masks in the caller's row order).
    SORTED_TRAINING = os.environ.get("CV_TRAIN_SORTED", "1") != "0"

    def modular_forward(self, x):
        if self.SORTED_TRAINING and x.tensor_stride == 1 and type(self.conv0p1s1) is ME.MinkowskiConvolution:
            cm_s, stem_map, out_map = x.coordinate_manager.fused_plan(self.conv0p1s1.kernel_size)
            n = x.F.shape[0]
            f0 = ME._ConvFn.apply(x.F, self.conv0p1s1.kernel, self.conv0p1s1.bias, stem_map, n)
            out = self._modular_body(ME.SparseTensor(f0, coordinate_manager=cm_s, tensor_stride=1))
            k = self.final.kernel
            y = ME._ConvFn.apply(out.F, k, self.final.bias, out_map, n)
            return x._like(y, 1)
        return self.final(self._modular_body(self.conv0p1s1(x)))

    def _modular_body(self, stem_out):
        out_p1 = self.bn0.forward_fused(stem_out, relu=True)
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
        return out

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

    # levels with at least this many rows run the mask-sorted grouped conv (the coordinate manager builds the orders)
    # None: follow ME.masked_min_rows() (the calling thread's launch policy, else the process-wide value); a number pins it for
    # this model outside a policy (profiles/sweep_mask_groups.py)
    MASKED_MIN_ROWS = None

    def masked_min_rows(self):
        v = getattr(ME._policy_tls, "masked_min_rows", None)
        if v is not None:
            return v
        return self.MASKED_MIN_ROWS if self.MASKED_MIN_ROWS is not None else ME.CoordinateManager.MASKED_MIN_ROWS
    MASK_GROUPS = ME.CoordinateManager.MASK_GROUPS
    # eval forward through the C executor (cv_net_run_f32, one call per scene: host time 1.46 -> 0.72 ms).  Off by
    # default: the GPU, not the host, bounds the scene rate today, and the executor's per-scene arena (257 MB at
    # 80k points, no aliasing across levels) costs more Infinity-Cache misses with several scenes in flight than
    # the caching allocator's recycled temporaries (227 vs 244 scenes/s at 3 scenes in flight).
    USE_PROGRAM = os.environ.get("CV_NET_PROGRAM", "1") != "0"
    # program mode: the 1x1 downsample conv of a block's first BasicBlock is folded into its conv2 (second source)
    FUSE_DOWNSAMPLE = os.environ.get("CV_FUSE_DOWNSAMPLE", "1") != "0"
    # program mode: fp32 products as three fp16 x fp16 piece products (operands split h + l, 22 significant bits and
    # the sign of l) instead of six bf16 ones.  fp16 has a range: a convolution whose input holds a magnitude above
    # 65000 raises a flag in pinned host memory and the forward is redone on the bf16 triples (check_range).
    PIECES = 2 if os.environ.get("CV_CONV_H2", "1") != "0" else 3
    # fp16-pair program: activations between the convolutions in the hl format (cv_conv_desc.in_hl)
    HL_BUFFERS = os.environ.get("CV_NET_HL", "1") != "0"
    # fp16-pair program: the 5x5x5 stem as a GEMM over the gathered operand on the matrix cores (conv_stem_mfma)
    STEM_MFMA = os.environ.get("CV_STEM_MFMA", "1") != "0"
    # False: forward() waits for the launches and checks the range flag itself; True: the caller does it after its
    # own synchronisation point (pipeline.detect_scene: no extra wait per scene)
    defer_range_check = False

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
        cm, stem_map, out_map = x.coordinate_manager.fused_plan(self.conv0p1s1.kernel_size)
        dev = x.F.device
        exp = self.BLOCK.expansion
        n = [cm.num_rows(1 << i) for i in range(5)]
        perms = lambda ts, rows: cm.mask_perms(3, ts, self.MASK_GROUPS) if rows >= self.masked_min_rows() else None
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


    # ------------------------------------------------------------------ the same forward as ONE C call per scene
    MAP_STEM, MAP_DOWN, MAP_K3, MAP_UP, MAP_OUT = 0, 1, 5, 10, 14          # slots of the per-scene map table
    PERM_K3, PERM_UP = 0, 5                                                 # slots of the processing-order table

    def _program(self, dev, pieces=3):
        """(ops, bufs, keep-alive) for cv_net_run_f32: the launch sequence of fused_forward with symbolic operands
        (arena buffer slots, map / order slots).  Built once per parameter version and piece format."""
        from . import _lib
        # cheap staleness check (walking nn.Module.parameters() costs 0.3 ms per call): the flat tensor list is
        # kept with the program, in-place updates bump _version, re-allocation (.to(), load_state_dict with
        # assign) changes data_ptr of the first parameter or the tensor objects themselves
        progs = self.__dict__.setdefault("_prog", {})
        hit = progs.get(pieces)
        if hit is not None:
            tensors, ver = hit[0]
            if ver == (sum(t._version for t in tensors), tensors[0].data_ptr(), str(dev), self.training):
                return hit[1]
        tensors = list(self.parameters()) + list(self.buffers())
        ver = ((tensors, (sum(t._version for t in tensors), tensors[0].data_ptr(), str(dev), self.training)))
        exp = self.BLOCK.expansion
        bufs, ops, keep, free = [], [], [], {}
        # fp16-pair program: every arena buffer holds the hl format (the pair of each value stored in place by the
        # producing epilogue; consumers load matrix-core fragments straight from it) - the caller's tensors stay fp32
        hl = 1 if (pieces == 2 and self.HL_BUFFERS and ME.CONV_X6) else 0

        def alloc(level, ch):
            pool = free.get((level, ch))
            if pool:
                return pool.pop()
            bufs.append((level, ch, level, hl))
            return len(bufs) - 1

        def release(slot):
            level, ch = bufs[slot][:2]
            if level >= 0:
                free.setdefault((level, ch), []).append(slot)

        def conv(src, dst, kernel, K, map_slot, scale=None, shift=None, res=None, relu=False, perm=-1, groups=0,
                 second=None):
            """second = (src2, kernel2 [cin2, cout], scale2): out += scale2 * (src2 @ kernel2) on the output rows; the
            BatchNorm scales are then folded into both packed weight sets and only `shift` remains in the epilogue"""
            w = (kernel if kernel.dim() == 3 else kernel[None]).detach().contiguous()
            in2 = (-1, 0)
            cin2, w6_2 = 0, None
            vec = ME.CONV_X6 and w.shape[1] % 32 == 0 and w.shape[2] % 4 == 0
            op_pieces, acc_scale = 3, 1.0
            if second is not None:
                src2, kernel2, scale2 = second
                w2 = (kernel2 if kernel2.dim() == 3 else kernel2[None]).detach().contiguous()
                if pieces == 1:
                    w6, w6_2 = ME.packed_weights_bf16(w, scale), ME.packed_weights_bf16(w2, scale2)
                    op_pieces = 1
                elif pieces == 2:
                    k = ME.h2_scale_log2((w, scale), (w2, scale2))           # one accumulator: one common factor
                    w6, w6_2 = ME.packed_weights_h2(w, scale, k), ME.packed_weights_h2(w2, scale2, k)
                    op_pieces, acc_scale = 2, 2.0 ** -k
                else:
                    w6 = ME.packed_weights_x6_scaled(w, scale)
                    w6_2 = ME.packed_weights_x6_scaled(w2, scale2)
                in2, cin2, scale = src2, w2.shape[1], None
            elif (pieces == 2 and hl and not vec and self.STEM_MFMA and w.shape[2] == 32 and w.shape[1] in (3, 6)
                  and w.shape[0] <= 128):
                # stem on the matrix cores: BatchNorm scale folded into the fp16-pair weights
                k = ME.h2_scale_log2((w, scale))
                w6 = ME.packed_weights_stem_h2(w, scale, k)
                op_pieces, acc_scale, scale = 2, 2.0 ** -k, None
            elif vec and pieces == 1:
                w6, op_pieces = ME.packed_weights_bf16(w), 1
            elif vec and pieces == 2:
                k = ME.h2_scale_log2((w, None))
                w6 = ME.packed_weights_h2(w, None, k)
                op_pieces, acc_scale = 2, 2.0 ** -k
            else:
                w6 = ME.packed_weights_x6(kernel, w) if vec else None
            keep.extend([w, scale, shift, w6, w6_2])
            ops.append(dict(in_buf=src[0], in_col=src[1], cin=w.shape[1], out_buf=dst[0], out_col=dst[1],
                            cout=w.shape[2], res_buf=res[0] if res else -1, res_col=res[1] if res else 0,
                            map=map_slot, K=K, perm=perm, perm_groups=groups, relu=1 if relu else 0,
                            weight=w.data_ptr(), scale=scale.data_ptr() if scale is not None else None,
                            shift=shift.data_ptr() if shift is not None else None,
                            weight_x6=w6.data_ptr() if w6 is not None else None,
                            in2_buf=in2[0], in2_col=in2[1], cin2=cin2,
                            weight2_x6=w6_2.data_ptr() if w6_2 is not None else None,
                            weight_pieces=op_pieces, acc_scale=acc_scale))

        def layer(seq, x, level, out_view):
            """x, out_view: (slot, first column); returns the (slot, column) holding the layer's output"""
            for bi, blk in enumerate(seq):
                planes = blk.conv1.out_channels
                s1, b1 = self._fold(blk.norm1)
                t = (alloc(level, planes), 0)
                k3 = dict(K=27, map_slot=self.MAP_K3 + level, perm=self.PERM_K3 + level, groups=self.MASK_GROUPS)
                conv(x, t, blk.conv1.kernel, scale=s1, shift=b1, relu=True, **k3)
                s2, b2 = self._fold(blk.norm2)
                last = bi == len(seq) - 1
                y = out_view if (last and out_view is not None) else (alloc(level, planes), 0)
                fuse_ds = (blk.downsample is not None and self.FUSE_DOWNSAMPLE and ME.CONV_X6 and
                           blk.conv1.in_channels % 32 == 0 and planes % 32 == 0)
                if fuse_ds:
                    # the 1x1 downsample branch rides in conv2 as a second source: one launch (and one split-K
                    # reduction) less per block; BatchNorm scales folded into the packed weights, shifts added
                    sd, bd = self._fold(blk.downsample[1])
                    shift = (b2 + bd).contiguous()
                    conv(t, y, blk.conv2.kernel, scale=s2, shift=shift, relu=True,
                         second=(x, blk.downsample[0].kernel, sd), **k3)
                    res = x
                else:
                    if blk.downsample is not None:
                        sd, bd = self._fold(blk.downsample[1])
                        res = (alloc(level, planes), 0)
                        conv(x, res, blk.downsample[0].kernel, 1, -1, scale=sd, shift=bd)
                    else:
                        res = x
                    conv(t, y, blk.conv2.kernel, scale=s2, shift=b2, res=res, relu=True, **k3)
                release(t[0])
                if res is not x:
                    release(res[0])
                if x[1] == 0 and bufs[x[0]][1] == (blk.conv1.in_channels):       # whole buffers only, never views
                    release(x[0])
                x = y
            return x

        bufs.append((-1, self.conv0p1s1.in_channels, 0, 0))       # slot 0: caller's features (original row order)
        bufs.append((-1, self.final.out_channels, 0, 0))          # slot 1: caller's output
        up_c = [self.PLANES[4 + i] for i in range(4)]
        skip_c = (self.PLANES[2] * exp, self.PLANES[1] * exp, self.PLANES[0] * exp, self.INIT_DIM)
        cat = []
        for i in range(4):
            bufs.append((3 - i, up_c[i] + skip_c[i], 3 - i, hl))
            cat.append(len(bufs) - 1)
        s, b = self._fold(self.bn0)
        out = (cat[3], up_c[3])
        conv((0, 0), out, self.conv0p1s1.kernel, self.conv0p1s1.kernel.shape[0], self.MAP_STEM, scale=s, shift=b,
             relu=True)
        for i, (cname, bname) in enumerate(_DOWN):
            s, b = self._fold(getattr(self, bname))
            c = getattr(self, cname)
            d = (alloc(i + 1, c.out_channels), 0)
            conv(out, d, c.kernel, 8, self.MAP_DOWN + i, scale=s, shift=b, relu=True)
            out = layer(getattr(self, "block%d" % (i + 1)), d, i + 1, (cat[2 - i], up_c[2 - i]) if i < 3 else None)
        for i, (cname, bname) in enumerate(_UP):
            lvl = 3 - i
            s, b = self._fold(getattr(self, bname))
            conv(out, (cat[i], 0), getattr(self, cname).kernel, 8, self.MAP_UP + i, scale=s, shift=b, relu=True,
                 perm=self.PERM_UP + i)
            if bufs[out[0]][0] >= 0 and out[1] == 0 and out[0] not in cat:
                release(out[0])
            out = layer(getattr(self, "block%d" % (5 + i)), (cat[i], 0), lvl, None)
        bias = self.final.bias.detach().reshape(-1).contiguous()
        conv(out, (1, 0), self.final.kernel, 1, self.MAP_OUT, shift=bias)
        c_ops = (_lib.NetOp * len(ops))(*[_lib.NetOp(**o) for o in ops])
        c_bufs = (_lib.NetBuf * len(bufs))(*[_lib.NetBuf(*bf) for bf in bufs])
        prog = (c_ops, c_bufs, keep)
        progs[pieces] = (ver, prog)
        return prog

    def check_range(self, x, y):
        """After the stream has been synchronised: if a convolution of the last fp16-pair forward on this stream
        saw an input beyond the fp16 range, redo the forward on the bf16 triples.  Returns the valid output."""
        flag = ME.range_flag(x.F.device)
        if int(flag[0]) == 0:
            return y
        flag.zero_()
        self.range_fallbacks = getattr(self, "range_fallbacks", 0) + 1
        return self.program_forward(x, pieces=3)

    def program_forward(self, x, pieces=None, defer_check=None):
        """fused_forward through the C executor (cv_net_run_f32): identical launches, one call."""
        import ctypes
        from . import _lib
        L = _lib.lib()
        dev = x.F.device
        if pieces is None:
            pieces = 1 if ME.COMPUTE_DTYPE == "bf16" else self.PIECES
        c_ops, c_bufs, _ = self._program(dev, pieces)
        plan = x.coordinate_manager.fused_fast(self.conv0p1s1.kernel_size)
        flag = ME.range_flag(dev) if pieces == 2 else None
        n = plan.counts
        masked = [n[i] >= self.masked_min_rows() for i in range(5)]
        if any(m and plan.perm_ptrs[i] is None for i, m in enumerate(masked)):
            # mask groups too wide for the plan's counting sort: orders from the generic path
            cm = x.coordinate_manager.fused_plan(self.conv0p1s1.kernel_size)[0]
            wide = [cm.mask_perms(3, 1 << i, self.MASK_GROUPS) if masked[i] else None for i in range(5)]
            perm_ptrs = [w.data_ptr() if w is not None else None for w in wide] + plan.perm_ptrs[5:]
        else:
            perm_ptrs = [p if masked[i] else None for i, p in enumerate(plan.perm_ptrs[:5])] + plan.perm_ptrs[5:]
        map_ptrs = plan.map_ptrs
        feats = x.F.contiguous()
        y = torch.empty((n[0], self.final.out_channels), dtype=torch.float32, device=dev)
        rows = (ctypes.c_int64 * 5)(*n)
        # activations of the executor: scratch of this stream (dead when `y` is written; sizes differ from scene to scene)
        arena = _lib.scratch(dev, "net_arena", L.cv_net_arena_bytes(c_bufs, len(c_bufs), rows, 5))
        cmax = max(self.PLANES)
        ws_bytes = max([4 * self.MASK_GROUPS * n[i] * cmax + 256 for i in range(5) if perm_ptrs[i] is not None] +
                       [int(L.cv_sp_conv_workspace_bytes(min(n[i], 128 * 384 - 1), cmax, 27)) for i in range(5)])
        ws = ME._workspace(dev, ws_bytes + 16384 + 256)      # + the split-K tickets the executor keeps at the tail
        vp = ctypes.c_void_p
        ext_ptr = (vp * 2)(feats.data_ptr(), y.data_ptr())
        ext_ld = (ctypes.c_int * 2)(feats.stride(0), y.stride(0))
        c_maps = (vp * len(map_ptrs))(*map_ptrs)
        c_perms = (vp * len(perm_ptrs))(*perm_ptrs)
        with torch.cuda.device(dev):
            _lib.check(L.cv_net_run_f32(c_ops, len(c_ops), c_bufs, len(c_bufs), rows, 5, vp(arena.data_ptr()),
                                        arena.numel(), ext_ptr, ext_ld, c_maps, len(map_ptrs), c_perms, len(perm_ptrs),
                                        vp(ws.data_ptr()), ws.numel(), vp(flag.data_ptr()) if flag is not None else None,
                                        vp(torch.cuda.current_stream(dev).cuda_stream)), "cv_net_run_f32")
        out = x._like(y, 1)
        if flag is not None and not (self.defer_range_check if defer_check is None else defer_check):
            torch.cuda.current_stream(dev).synchronize()
            out = self.check_range(x, out)
        return out

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
