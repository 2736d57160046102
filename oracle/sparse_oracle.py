"""CPU oracle of the sparse-voxel network -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, with torch CPU fp32 gather-matmul-scatter, the subset of MinkowskiEngine 0.5.3
that the reference network uses, and the reference's own topology:

  MinkUNet34C forward      <- utils/minkunet.py:50-120 (modules), :122-180 (forward),
                              :193-195,:244-245 (LAYERS / PLANES of MinkUNet34C)
  _make_layer / BasicBlock <- utils/resnet.py:118-154 (1x1 conv + BN downsample iff channels
                              change); BasicBlock is ME's (conv1,norm1,relu,conv2,norm2,+res,relu)
  head split (eval)        <- eval_joint.py:173-190

PARITY STATUS: the COMPOSITION (layer order, widths, strides, concatenations, residual adds,
BN/ReLU placement, parameter names and shapes) and the head split are PINNED BY REFERENCE
EXECUTION: tests/golden/make_net_golden.py imports the reference's own MinkUNet34C class and
runs its forward with the MinkowskiEngine names bound to the primitives below
(tests/golden/net_ref.npz); make_decode_golden.py exec()s eval_joint.py:173-190 for the head.
The PRIMITIVE ARITHMETIC is "parity unpinned": MinkowskiEngine v0.5.3 (README.md:53) is an external
dependency that is neither vendored in the reference nor installed here, and the reference
has no test at this boundary.  The [ME-ext] semantics below are the published algorithm
(Choy et al., 4D Spatio-Temporal ConvNets, generalized sparse convolution):
  out[u] = sum_{o in K} W_o^T x[u*s + o*ts]        over ACTIVE inputs only
  - odd kernels are centred, even kernels use offsets 0..k-1
  - stride-2 output coordinates = unique(floor(c / (2 ts)) * 2 ts)
  - transposed k2s2 conv writes onto the existing finer coordinate set:
    out[v] = W_{oct(v)}^T x[parent(v)]
  - weight `kernel` is [K^3, Cin, Cout] ([Cin, Cout] when K = 1), kernel-offset index runs
    with the FIRST spatial axis fastest (KERNEL_OFFSET_ORDER is the one place to flip it
    when a real checkpoint is available)
  - MinkowskiBatchNorm = nn.BatchNorm1d over the [N, C] feature rows (eps 1e-5)
It is pinned here by tests/test_sparse_oracle.py against dense torch.nn.functional
conv3d / conv_transpose3d / batch_norm on densified grids masked to the active set.
"""
import numpy as np
import torch
import torch.nn.functional as F

KERNEL_OFFSET_ORDER = "x_fastest"

LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)              # utils/minkunet.py:193-195 (MinkUNet34)
PLANES = (32, 64, 128, 256, 256, 128, 96, 96)  # utils/minkunet.py:244-245 (MinkUNet34C)
INIT_DIM = 32


def kernel_offsets(k):
    """[K^3, 3] integer offsets (units of the input tensor stride), kernel-index order."""
    rng = np.arange(k) - k // 2 if k % 2 == 1 else np.arange(k)
    if KERNEL_OFFSET_ORDER == "x_fastest":
        zz, yy, xx = np.meshgrid(rng, rng, rng, indexing="ij")
    else:
        xx, yy, zz = np.meshgrid(rng, rng, rng, indexing="ij")
    return np.stack([xx.ravel(), yy.ravel(), zz.ravel()], -1).astype(np.int64)


def _keys(coords):
    """[N,4] (b,x,y,z) int -> unique int64 key"""
    c = np.asarray(coords, np.int64)
    return ((c[:, 0] << 48) | ((c[:, 1] + 32768) << 32) | ((c[:, 2] + 32768) << 16) | (c[:, 3] + 32768))


def downsample_coords(coords, ts):
    """unique(floor(c / (2ts)) * 2ts) in order of first appearance."""
    c = np.asarray(coords, np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], 2 * ts) * (2 * ts)
    _, first = np.unique(_keys(c), return_index=True)
    return c[np.sort(first)]


def _lookup(keys_sorted, order, q):
    pos = np.searchsorted(keys_sorted, q)
    pos = np.clip(pos, 0, len(keys_sorted) - 1)
    hit = keys_sorted[pos] == q
    return np.where(hit, order[pos], -1)


def kernel_map(in_coords, out_coords, k, ts_in, stride):
    """nbr[N_out, K^3]: index into in_coords of out_coord + offset*ts_in, or -1."""
    kin = _keys(in_coords)
    order = np.argsort(kin, kind="stable")
    ks = kin[order]
    offs = kernel_offsets(k) * ts_in
    oc = np.asarray(out_coords, np.int64)
    nbr = np.empty((len(oc), len(offs)), np.int64)
    for j, o in enumerate(offs):
        q = oc.copy()
        q[:, 1:] += o[None]
        nbr[:, j] = _lookup(ks, order, _keys(q))
    return nbr


def conv(x, kernel, nbr, bias=None):
    """x [N_in,Cin], kernel [K,Cin,Cout] or [Cin,Cout], nbr [N_out,K] -> [N_out,Cout]"""
    if kernel.dim() == 2:
        kernel = kernel[None]
    out = torch.zeros((nbr.shape[0], kernel.shape[2]), dtype=x.dtype)
    for j in range(kernel.shape[0]):
        src = nbr[:, j]
        sel = np.nonzero(src >= 0)[0]
        if sel.size:
            out[sel] += x[src[sel]] @ kernel[j]
    if bias is not None:
        out = out + bias.reshape(1, -1)
    return out


def conv_transpose_k2s2(x, kernel, nbr_down):
    """x at the coarse set, nbr_down [N_coarse, 8] (map of the matching k2s2 conv: coarse <- fine).
    Output on the fine set: out[fine] = W_o^T x[coarse] for the (coarse, o) that contains it."""
    n_fine = int(nbr_down.max()) + 1
    out = torch.zeros((n_fine, kernel.shape[2]), dtype=x.dtype)
    for j in range(kernel.shape[0]):
        fine = nbr_down[:, j]
        sel = np.nonzero(fine >= 0)[0]
        if sel.size:
            out[fine[sel]] = x[sel] @ kernel[j]
    return out


# Test hook: an iterator of boolean masks, one per ReLU of the forward in call order (or None).  With it the k-th ReLU is
# x * mask_k instead of max(x, 0): a second evaluation (another precision, the HIP path) can be given the FIRST one's
# activation pattern, so that gradients are compared on the same piecewise-linear branch - pre-activations within
# rounding of zero otherwise pick different branches and change a gradient element by its whole value
# (tests/test_production_size_gpu.py).  relu_trace, when a list, receives every ReLU's mask (x > 0); relu_peaks, when a
# list, the largest value every ReLU lets through (the inputs of the next convolution: tests/test_train_gpu.py checks
# them against the fp16 range on trained weights).
relu_masks = None
relu_trace = None
relu_peaks = None


def _relu(x):
    if relu_masks is not None:
        m = next(relu_masks)
        y = x * torch.as_tensor(m).to(x.dtype)
    else:
        y = torch.relu(x)
    if relu_trace is not None:
        relu_trace.append((y.detach() > 0))
    if relu_peaks is not None:
        relu_peaks.append(float(y.detach().abs().max()) if y.numel() else 0.0)
    return y


def batch_norm(x, sd, prefix, training=False, eps=1e-5):
    return F.batch_norm(x, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"],
                        sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], training=training,
                        momentum=0.1, eps=eps)


class CoordinateManager:
    """Coordinate sets per tensor stride and the 10 kernel maps of one forward."""

    def __init__(self, coords):
        self.coords = {1: np.asarray(coords, np.int64)}
        for ts in (1, 2, 4, 8):
            self.coords[2 * ts] = downsample_coords(self.coords[ts], ts)
        self.maps = {}

    def map(self, k, ts, stride=1):
        key = (k, ts, stride)
        if key not in self.maps:
            out = self.coords[ts * stride]
            self.maps[key] = kernel_map(self.coords[ts], out, k, ts, stride)
        return self.maps[key]


def _block(x, sd, name, cm, ts, training):
    """ME BasicBlock (expansion 1) with the optional downsample of resnet.py:126-134."""
    nbr = cm.map(3, ts)
    out = conv(x, sd[name + ".conv1.kernel"], nbr)
    out = _relu(batch_norm(out, sd, name + ".norm1", training))
    out = conv(out, sd[name + ".conv2.kernel"], nbr)
    out = batch_norm(out, sd, name + ".norm2", training)
    if name + ".downsample.0.kernel" in sd:
        res = conv(x, sd[name + ".downsample.0.kernel"], cm.map(1, ts))
        res = batch_norm(res, sd, name + ".downsample.1", training)
    else:
        res = x
    return _relu(out + res)


def _layer(x, sd, name, n, cm, ts, training):
    for i in range(n):
        x = _block(x, sd, "%s.%d" % (name, i), cm, ts, training)
    return x


def minkunet34c_forward(sd, coords, feats, training=False, return_intermediates=False, dtype=torch.float32):
    """utils/minkunet.py:122-180 on (coords [N,4] int (b,x,y,z), feats [N,Cin]) -> [N, Cout].
    dtype=torch.float64 runs the same graph in double precision: the yardstick that says how far ANY fp32 evaluation
    (this oracle's or the HIP path's) sits from the exact result - ReLU masks of values within rounding of zero flip
    between two fp32 evaluations, and a flipped mask changes a gradient element by its whole value."""
    # tensors that require grad are used as they are so torch autograd can differentiate the oracle
    sd = {k: (v if (torch.is_tensor(v) and v.requires_grad) else
              (v.detach().to(dtype if v.dtype.is_floating_point else v.dtype).cpu().clone() if torch.is_tensor(v) else v))
          for k, v in sd.items()}
    cm = CoordinateManager(coords)
    x = torch.as_tensor(feats, dtype=dtype)
    inter = {}

    def cbr(x, conv_name, bn_name, k, ts, stride=1):
        y = conv(x, sd[conv_name + ".kernel"], cm.map(k, ts, stride))
        return _relu(batch_norm(y, sd, bn_name, training))

    def up(x, conv_name, bn_name, ts_coarse):
        y = conv_transpose_k2s2(x, sd[conv_name + ".kernel"], cm.map(2, ts_coarse // 2, 2))
        return _relu(batch_norm(y, sd, bn_name, training))

    out_p1 = cbr(x, "conv0p1s1", "bn0", 5, 1)                                    # :123-125
    out = cbr(out_p1, "conv1p1s2", "bn1", 2, 1, 2)                               # :127-129
    out_b1p2 = _layer(out, sd, "block1", LAYERS[0], cm, 2, training)             # :130
    out = cbr(out_b1p2, "conv2p2s2", "bn2", 2, 2, 2)
    out_b2p4 = _layer(out, sd, "block2", LAYERS[1], cm, 4, training)
    out = cbr(out_b2p4, "conv3p4s2", "bn3", 2, 4, 2)
    out_b3p8 = _layer(out, sd, "block3", LAYERS[2], cm, 8, training)
    out = cbr(out_b3p8, "conv4p8s2", "bn4", 2, 8, 2)                             # :147-149
    out = _layer(out, sd, "block4", LAYERS[3], cm, 16, training)
    inter["block4"] = out
    out = up(out, "convtr4p16s2", "bntr4", 16)                                   # :153-155
    out = _layer(torch.cat([out, out_b3p8], 1), sd, "block5", LAYERS[4], cm, 8, training)
    out = up(out, "convtr5p8s2", "bntr5", 8)
    out = _layer(torch.cat([out, out_b2p4], 1), sd, "block6", LAYERS[5], cm, 4, training)
    out = up(out, "convtr6p4s2", "bntr6", 4)
    out = _layer(torch.cat([out, out_b1p2], 1), sd, "block7", LAYERS[6], cm, 2, training)
    out = up(out, "convtr7p2s2", "bntr7", 2)
    out = _layer(torch.cat([out, out_p1], 1), sd, "block8", LAYERS[7], cm, 1, training)
    inter["block8"] = out
    y = conv(out, sd["final.kernel"], cm.map(1, 1), sd["final.bias"])            # :180
    if return_intermediates:
        inter["out_p1"] = out_p1
        inter["out_b1p2"] = out_b1p2
        return y, inter, cm
    return y


def head_joint_eval(out, nclasses=9, log_scale=True):
    """eval_joint.py:173-190: per-point head select by argmax class."""
    out = torch.as_tensor(out)
    o_xyz = out[:, :3 * nclasses]
    o_scale = out[:, 3 * nclasses:6 * nclasses]
    o_class = out[:, 6 * nclasses:]
    idx = o_class.argmax(-1)
    idx = torch.where(idx == nclasses, torch.zeros_like(idx), idx)
    g = idx[:, None, None].expand(-1, 1, 3)
    xyz = torch.gather(o_xyz.reshape(-1, nclasses, 3), 1, g)[:, 0]
    scale = torch.gather(o_scale.reshape(-1, nclasses, 3), 1, g)[:, 0]
    if log_scale:
        scale = torch.exp(scale)
    cls = torch.argmax(o_class[..., :-1], dim=-1)
    prob = torch.max(torch.softmax(o_class, dim=-1)[..., :-1], dim=-1)[0]
    return xyz, scale, prob, cls


def head_separate_eval(out, log_scale=True):
    """eval_separate.py:170-181"""
    out = torch.as_tensor(out)
    scale = torch.exp(out[:, 3:6]) if log_scale else out[:, 3:6]
    return out[:, :3], scale, torch.softmax(out[:, 6:8], dim=-1)[:, 1]


def make_state_dict(in_channels=3, out_channels=64, seed=0):
    """Random MinkUNet34C parameters with the reference's names and init
    (utils/resnet.py:109-116: kaiming-normal fan_out/relu on conv kernels, BN gamma 1 beta 0;
    transposed convs keep a uniform default).  BN running stats are randomised so that eval
    mode is a non-trivial affine map."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv_w(name, k, cin, cout, transposed=False):
        K = k ** 3
        shape = (cin, cout) if K == 1 else (K, cin, cout)
        if transposed:
            bound = 1.0 / np.sqrt(cin * K)
            w = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            std = np.sqrt(2.0 / (cout * K))          # fan_out = Cout * K^3
            w = torch.randn(shape, generator=g) * std
        sd[name + ".kernel"] = w

    def bn(name, c):
        sd[name + ".bn.weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bn.bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".bn.running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".bn.running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[name + ".bn.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    inpl = INIT_DIM
    conv_w("conv0p1s1", 5, in_channels, inpl); bn("bn0", inpl)

    def layer(name, planes, n, inplanes):
        for i in range(n):
            p = "%s.%d" % (name, i)
            cin = inplanes if i == 0 else planes
            conv_w(p + ".conv1", 3, cin, planes); bn(p + ".norm1", planes)
            conv_w(p + ".conv2", 3, planes, planes); bn(p + ".norm2", planes)
            if i == 0 and cin != planes:
                conv_w(p + ".downsample.0", 1, cin, planes); bn(p + ".downsample.1", planes)
        return planes

    for lvl, (cname, bname) in enumerate((("conv1p1s2", "bn1"), ("conv2p2s2", "bn2"),
                                          ("conv3p4s2", "bn3"), ("conv4p8s2", "bn4"))):
        conv_w(cname, 2, inpl, inpl); bn(bname, inpl)
        inpl = layer("block%d" % (lvl + 1), PLANES[lvl], LAYERS[lvl], inpl)
    skips = (PLANES[2], PLANES[1], PLANES[0], INIT_DIM)
    for j, (cname, bname) in enumerate((("convtr4p16s2", "bntr4"), ("convtr5p8s2", "bntr5"),
                                        ("convtr6p4s2", "bntr6"), ("convtr7p2s2", "bntr7"))):
        conv_w(cname, 2, inpl, PLANES[4 + j], transposed=True); bn(bname, PLANES[4 + j])
        inpl = layer("block%d" % (5 + j), PLANES[4 + j], LAYERS[4 + j], PLANES[4 + j] + skips[j])
    conv_w("final", 1, PLANES[7], out_channels)
    sd["final.bias"] = 0.1 * torch.randn(1, out_channels, generator=g)
    return sd
