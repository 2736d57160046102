"""MinkowskiEngine-compatible facade over the gfx950 sparse-voxel engine (libcvhip.so).

Only the subset the reference touches: ``SparseTensor`` (``.F``, ``.C``,
``decomposed_coordinates_and_features``), ``MinkowskiConvolution``,
``MinkowskiConvolutionTranspose``, ``MinkowskiBatchNorm`` (sub-module ``.bn``),
``MinkowskiReLU``, ``cat``, ``modules.resnet_block.BasicBlock``,
``utils.batched_coordinates / sparse_quantize / kaiming_normal_``.
Call sites: utils/minkunet.py:53-180, utils/resnet.py:109-154, train_joint.py:82,250,
eval_joint.py:64,169, utils/dataloader.py:197.

Parameter names and shapes follow MinkowskiEngine 0.5.x so reference checkpoints keep their
keys: conv weight ``kernel`` [K^3, Cin, Cout] ([Cin, Cout] when K = 1), ``bias`` [1, Cout],
``MinkowskiBatchNorm.bn`` = ``nn.BatchNorm1d``.

Row-order invariant relied on by the reference (train_joint.py:256-272): row i of a
stride-1 output corresponds to row i of the input coordinates.
"""
import ctypes
import math
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from . import utils  # noqa: F401

__version__ = "0.5.3-cvhip"


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _stream(dev):
    # (the raw handle: torch.cuda.current_stream builds a Stream object per call - 4 us x ~10 per layer of a training step)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device()))


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev):
    """`with torch.cuda.device(dev)` that costs nothing when dev is the current device already (one process per GPU: always)"""
    if dev.index is None or torch.cuda.current_device() == dev.index:
        return _NO_SWITCH
    return torch.cuda.device(dev)


class _FusedPlan:
    """what CoordinateManager.fused_fast() returns"""
    __slots__ = ("keep", "counts", "cap", "stem_k", "groups", "off", "layout", "map_ptrs", "perm_ptrs", "views")


class CoordinateManager:
    """Coordinate sets per tensor stride, their hash tables and the cached kernel maps."""

    NUM_LEVELS = 5

    def __init__(self, coords, num_levels=NUM_LEVELS, check=True, lazy=False):
        """check=False with num_levels=1 builds the level-0 hash without any host sync (the
        duplicate count stays on the device).  lazy=True (what SparseTensor passes): nothing is built until a map,
        a coarser level or the coordinates of this (caller-order) set are asked for - the fused network never does,
        it runs on the spatially sorted twin from fused_plan()."""
        assert coords.is_cuda and coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        self.device = coords.device
        if coords.shape[0] == 0:
            raise RuntimeError("SparseTensor: empty coordinate set")
        self.cap = int(_lib.lib().cv_sp_table_capacity(coords.shape[0]))
        self._input = coords.contiguous()
        self._fused = None
        self._lazy = (num_levels, check) if lazy else None
        if not lazy:
            self._build(num_levels, check)

    def __getattr__(self, name):
        # attributes created by _build(): a lazy manager builds on first use
        if name in ("coords", "counts", "_level", "_keys", "_vals", "_coords_buf", "_counts_d", "_verified", "_maps",
                    "num_levels") and self.__dict__.get("_lazy") is not None:
            lazy, self._lazy = self._lazy, None
            self._build(*lazy)
            return self.__dict__[name]
        raise AttributeError(name)

    @classmethod
    def _from_levels(cls, coords_buf, keys, vals, counts_d, counts, cap):
        """manager over levels built elsewhere (cv_sp_scene_plan)"""
        self = object.__new__(cls)
        self.device = coords_buf[0].device
        self.cap = cap
        self._input = coords_buf[0]
        self._fused = None
        self._lazy = None
        self.num_levels = len(coords_buf)
        self._coords_buf, self._keys, self._vals, self._counts_d = coords_buf, keys, vals, counts_d
        self._verified = True
        self.counts = counts
        self.coords = {1 << i: coords_buf[i][:counts[i]] for i in range(self.num_levels)}
        self._level = {1 << i: i for i in range(self.num_levels)}
        self._maps = {}
        return self

    def _build(self, num_levels, check):
        L = _lib.lib()
        dev = self.device
        n = self._input.shape[0]
        self.num_levels = num_levels
        self._coords_buf = [self._input] + [torch.empty((n, 4), dtype=torch.int32, device=dev)
                                            for _ in range(num_levels - 1)]
        self._keys = [torch.empty(self.cap, dtype=torch.int64, device=dev) for _ in range(num_levels)]
        self._vals = [torch.empty(self.cap, dtype=torch.int32, device=dev) for _ in range(num_levels)]
        self._counts_d = torch.empty(8, dtype=torch.int32, device=dev)
        counts_h = (ctypes.c_int32 * 8)()
        ws = torch.empty(int(L.cv_sp_levels_workspace_bytes(n)), dtype=torch.uint8, device=dev)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        sync = check or num_levels > 1
        with _on(dev):
            _lib.check(L.cv_sp_build_levels(arr(self._coords_buf), arr(self._keys), arr(self._vals), n,
                                            self.cap, num_levels, _ptr(self._counts_d),
                                            counts_h if sync else None, _ptr(ws), ws.numel(), _stream(dev)),
                       "cv_sp_build_levels")
        self._verified = sync
        if sync:
            self._raise_on_dups(counts_h[5], counts_h[6])
            self.counts = [int(counts_h[i]) for i in range(num_levels)]
        else:
            self.counts = [n]
        self.coords = {1 << i: self._coords_buf[i][:self.counts[i]] for i in range(num_levels)}
        self._level = {1 << i: i for i in range(num_levels)}
        self._maps = {}

    @staticmethod
    def _raise_on_dups(count, out_of_range=0):
        if out_of_range != 0:
            raise RuntimeError("SparseTensor: %d coordinates outside the supported window (spatial coordinates in "
                               "[-32704, 32703], batch index < 65536)" % out_of_range)
        if count != 0:
            raise RuntimeError("SparseTensor: %d duplicate coordinates (quantise with "
                               "utils.sparse_quantize first, as utils/dataloader.py:197 does)" % count)

    def _verify(self):
        """deferred duplicate check of a set built without a host sync (first map request pays it)"""
        if not self._verified:
            c = self._counts_d[5:7].tolist()
            self._raise_on_dups(int(c[0]), int(c[1]))
            self._verified = True

    def ensure_levels(self, num_levels=NUM_LEVELS):
        if self.num_levels < num_levels:
            self._build(num_levels, True)

    def cross_map(self, out_coords, k, ts=1):
        """kernel map of an arbitrary output set looked up in THIS manager's level-0 table"""
        L = _lib.lib()
        m = torch.empty((out_coords.shape[0], k ** 3), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.cv_sp_kernel_map(_ptr(out_coords), out_coords.shape[0], _ptr(self._keys[0]),
                                          _ptr(self._vals[0]), self.cap, k, ts, _ptr(m), _stream(self.device)),
                       "cv_sp_kernel_map")
        return m

    def mask_perms(self, k, ts, groups=2):
        """int32 [groups, n]: for each contiguous group of the k^3 offsets, the row order sorted by the
        bit mask of valid neighbours within that group (stride-1 map at tensor stride ts, cached)."""
        key = ("mp", k, ts, groups)
        m = self._maps.get(key)
        if m is None:
            L = _lib.lib()
            nbr = self.kernel_map(k, ts)
            n, K = nbr.shape
            if (K + groups - 1) // groups <= 10:
                m = _perms_with_map(nbr, groups)
            else:       # wide groups: generic key + device sort
                m = torch.empty((groups, n), dtype=torch.int32, device=self.device)
                keys = torch.empty(n, dtype=torch.int64, device=self.device)
                for g in range(groups):
                    jb, je = K * g // groups, K * (g + 1) // groups      # same split as the kernel
                    with torch.cuda.device(self.device):
                        _lib.check(L.cv_sp_mask_keys(_ptr(nbr), n, K, jb, je, _ptr(keys), _stream(self.device)),
                                   "cv_sp_mask_keys")
                    m[g] = torch.argsort(keys).to(torch.int32)
            self._maps[key] = m
        return m

    def up_perm(self, ts_coarse):
        """row order of the fine set sorted by octant (the 8-bit mask of the transposed-conv map): tiles of
        128 rows then share one kernel offset and the other seven are skipped."""
        key = ("upperm", ts_coarse)
        m = self._maps.get(key)
        if m is None:
            L = _lib.lib()
            up = self.up_map(ts_coarse)
            m = torch.empty((1, up.shape[0]), dtype=torch.int32, device=self.device)
            ws = torch.empty(4096, dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.cv_sp_mask_perms(_ptr(up), up.shape[0], 8, 1, _ptr(m), _ptr(ws), ws.numel(), 0,
                                              _stream(self.device)), "cv_sp_mask_perms")
            m = m[0]
            self._maps[key] = m
        return m

    # mask-sorted offset groups of the 3x3x3 convs: levels with at least MASKED_MIN_ROWS rows get MASK_GROUPS orders.
    # Three groups since the hl-format kernels (round 2): 3 / 4 / 5 groups = net 2.44 / 2.47 / 2.53 ms, 507 / 486 / 476
    # scenes/s six in flight - a quarter less partial-sum traffic is worth more than the 4 % more MFMA blocks (four groups
    # were better with the round-1 kernels: 2.99 vs 3.20 ms)
    MASK_GROUPS = int(os.environ.get("CV_NET_MASK_GROUPS", "3"))
    MASKED_MIN_ROWS = int(os.environ.get("CV_MASKED_MIN_ROWS", "16384"))      # process-wide; a thread's own: masked_min_rows()
    LIB_MASKED_MIN_ROWS = MASKED_MIN_ROWS          # the one-scene-at-a-time default (pipeline.policy_for_scenes_in_flight)

    def fused_fast(self, stem_k=5):
        """The coordinate plan of the fused network as raw device pointers (what MinkUNet.program_forward hands to the
        C executor): .counts rows per level, .map_ptrs [stem, down 0-3, k3 0-4, up 0-3, out], .perm_ptrs [mask orders of
        levels 0-4 (None where a level is not mask-sorted), octant orders of the four transposed convs].
        ONE C call (cv_sp_scene_plan: spatial row sort, the five levels of the sorted set, every map and order) into
        three allocations; the tensor views of the plan (fused_plan) are only made when somebody asks for them."""
        cache = self.__dict__.setdefault("_fused_cache", {})
        mmr = masked_min_rows()                # (part of the key: a plan built under another launch policy has other orders)
        if (stem_k, mmr) in cache:
            return cache[(stem_k, mmr)]
        L = _lib.lib()
        dev = self.device
        n = self._input.shape[0]
        NL = CoordinateManager.NUM_LEVELS
        G = self.MASK_GROUPS if (27 + self.MASK_GROUPS - 1) // self.MASK_GROUPS <= 10 else 0   # wide groups: lazily
        cap = int(L.cv_sp_table_capacity(n))
        words = int(L.cv_sp_scene_plan_words(n, stem_k, G, mmr))
        up64 = lambda v: (v + 63) // 64 * 64
        # int32 buffer: perm | inv | coords of the 5 levels | table values of the 5 levels | counts | arena (sized for
        # the worst case, every coarse level bounded by n: the call does not come back between the levels and the maps)
        o_perm, o_inv = 0, up64(n)
        o_coords = [o_inv + up64(n) + i * up64(4 * n) for i in range(NL)]
        o_vals = [o_coords[-1] + up64(4 * n) + i * up64(cap) for i in range(NL)]
        o_counts = o_vals[-1] + up64(cap)
        o_arena = o_counts + 64
        ibuf = torch.empty(o_arena + words, dtype=torch.int32, device=dev)
        kbuf = torch.empty(NL * cap, dtype=torch.int64, device=dev)
        sws_b, lws_b = int(L.cv_sp_sort_workspace_bytes(n)), int(L.cv_sp_levels_workspace_bytes(n))
        wbuf = _lib.scratch(dev, "scene_plan", up64(sws_b) + lws_b)      # sort + level workspaces: dead when the call returns
        ib, kb, wb = ibuf.data_ptr(), kbuf.data_ptr(), wbuf.data_ptr()
        vp = ctypes.c_void_p
        c_coords = (vp * NL)(*[ib + 4 * o for o in o_coords])
        c_keys = (vp * NL)(*[kb + 8 * cap * i for i in range(NL)])
        c_vals = (vp * NL)(*[ib + 4 * o for o in o_vals])
        counts_h = (ctypes.c_int32 * 8)()
        off = _lib.SceneMaps()
        with _on(dev):
            _lib.check(L.cv_sp_scene_plan(_ptr(self._input), n, vp(ib + 4 * o_perm), vp(ib + 4 * o_inv), c_coords, c_keys,
                                          c_vals, cap, vp(ib + 4 * o_counts), counts_h, stem_k, G, mmr,
                                          vp(ib + 4 * o_arena), words, ctypes.byref(off), vp(wb), sws_b, vp(wb + up64(sws_b)),
                                          lws_b, _stream(dev)), "cv_sp_scene_plan")
        self._raise_on_dups(counts_h[5], counts_h[6])
        plan = _FusedPlan()
        plan.keep = (ibuf, kbuf)
        plan.counts = [int(counts_h[i]) for i in range(NL)]
        plan.cap, plan.stem_k, plan.groups, plan.off = cap, stem_k, G, off
        plan.layout = (o_perm, o_inv, o_coords, o_vals, o_counts, o_arena)
        ap = lambda o: ib + 4 * (o_arena + o)
        plan.map_ptrs = [ap(off.stem)] + [ap(off.down[i]) for i in range(4)] + [ap(off.k3[i]) for i in range(5)] + \
                        [ap(off.up[i]) for i in range(4)] + [ib + 4 * o_inv]
        plan.perm_ptrs = [ap(off.mask_perm[i]) if off.mask_perm[i] >= 0 else None for i in range(5)] + \
                         [ap(off.up_perm[i]) for i in range(4)]
        plan.views = None
        cache[(stem_k, mmr)] = plan
        self._fused = plan
        self._fused_k = stem_k
        return plan

    def fused_plan(self, stem_k=5):
        """Spatially sorted twin of this coordinate set for the fused network:
        (sorted manager with all levels, stem map sorted<-original rows, final map original<-sorted) as tensors -
        views of the buffers fused_fast() filled, every kernel map / processing order pre-seeded into the sorted
        manager's cache."""
        plan = self.fused_fast(stem_k)
        if plan.views is None:
            ibuf, kbuf = plan.keep
            n, cap, G, off = self._input.shape[0], plan.cap, plan.groups, plan.off
            o_perm, o_inv, o_coords, o_vals, o_counts, o_arena = plan.layout
            c = plan.counts
            NL = len(c)
            coords_buf = [ibuf[o:o + 4 * n].view(n, 4) for o in o_coords]
            keys = [kbuf[i * cap:(i + 1) * cap] for i in range(NL)]
            vals = [ibuf[o:o + cap] for o in o_vals]
            cm_s = CoordinateManager._from_levels(coords_buf, keys, vals, ibuf[o_counts:o_counts + 8], c, cap)
            arena = ibuf[o_arena:]
            view = lambda o, r, k: arena[o:o + r * k].view(r, k)
            stem_map = view(off.stem, c[0], stem_k ** 3)
            out_map = ibuf[o_inv:o_inv + n].view(n, 1)   # original row <- sorted row: the final 1x1 conv's "kernel map"
            for i in range(4):
                cm_s._maps[("k", 2, 1 << i, 2)] = view(off.down[i], c[i + 1], 8)
                cm_s._maps[("up", 16 >> i)] = view(off.up[i], c[3 - i], 8)
                cm_s._maps[("upperm", 16 >> i)] = arena[off.up_perm[i]:off.up_perm[i] + c[3 - i]]
            for i in range(5):
                cm_s._maps[("k", 3, 1 << i, 1)] = view(off.k3[i], c[i], 27)
                if off.mask_perm[i] >= 0:
                    mp = view(off.mask_perm[i], G, c[i])
                    mp._cv_has_map = True              # the map rows in processing order follow in the arena
                    mp._cv_from_plan = True
                    cm_s._maps[("mp", 3, 1 << i, G)] = mp
                    cm_s._maps[("k", 3, 1 << i, 1)]._cv_mask_perms = mp      # the generic conv path finds the orders on the map
            plan.views = (cm_s, stem_map, out_map)
        return plan.views

    def num_rows(self, ts):
        if ts == 1:
            return self._input.shape[0]
        if ts not in self._level:
            self.ensure_levels()
        return self.counts[self._level[ts]]

    def kernel_map(self, k, ts_in, stride=1):
        """int32 [n_out, k^3]: rows of the ts_in set feeding each row of the ts_in*stride set."""
        if ts_in * stride not in self._level:
            self.ensure_levels()
        self._verify()
        key = ("k", k, ts_in, stride)
        m = self._maps.get(key)
        if m is None:
            L = _lib.lib()
            li = self._level[ts_in]
            out = self.coords[ts_in * stride]
            m = torch.empty((out.shape[0], k ** 3), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.cv_sp_kernel_map(_ptr(out), out.shape[0], _ptr(self._keys[li]),
                                              _ptr(self._vals[li]), self.cap, k, ts_in, _ptr(m),
                                              _stream(self.device)), "cv_sp_kernel_map")
            self._maps[key] = m
        return m

    def up_map(self, ts_coarse):
        """int32 [n_fine, 8] map of the transposed k2s2 conv ts_coarse -> ts_coarse/2."""
        key = ("up", ts_coarse)
        m = self._maps.get(key)
        if m is None:
            L = _lib.lib()
            down = self.kernel_map(2, ts_coarse // 2, 2)
            n_fine = self.num_rows(ts_coarse // 2)
            m = torch.empty((n_fine, 8), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(L.cv_sp_up_map(_ptr(down), down.shape[0], n_fine, _ptr(m), _stream(self.device)),
                           "cv_sp_up_map")
            self._maps[key] = m
        return m


class SparseTensor:
    """``ME.SparseTensor(features, coordinates, device=...)`` (train_joint.py:250)."""

    def __init__(self, features, coordinates=None, device=None, coordinate_manager=None, tensor_stride=1,
                 **_):
        if device is None:
            device = features.device if coordinates is None else (
                features.device if features.is_cuda else torch.device("cuda"))
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("canonicalvoting_amd SparseTensor runs on the GPU only (no CPU engine); "
                               "pass device='cuda'")
        self.F = features.to(device=device, dtype=torch.float32)
        self.F_hl = None        # training forward: the same rows in the hl format (written by the producing BatchNorm pass)
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("coordinates or coordinate_manager required")
            c = coordinates
            if c.is_floating_point():
                c = torch.floor(c)
            coordinate_manager = CoordinateManager(c.to(device=device, dtype=torch.int32).contiguous(),
                                                   num_levels=1, check=False, lazy=True)
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = tensor_stride if isinstance(tensor_stride, int) else int(tensor_stride[0])
        if self.F.shape[0] != coordinate_manager.num_rows(self.tensor_stride):
            raise RuntimeError("features have %d rows, coordinate set has %d" % (
                self.F.shape[0], coordinate_manager.num_rows(self.tensor_stride)))

    @property
    def C(self):
        cm = self.coordinate_manager
        if self.tensor_stride not in cm.coords:
            cm.ensure_levels()
        return cm.coords[self.tensor_stride]

    @property
    def coordinates(self):
        return self.C

    @property
    def features(self):
        return self.F

    @property
    def device(self):
        return self.F.device

    @property
    def shape(self):
        return self.F.shape

    def __len__(self):
        return self.F.shape[0]

    @property
    def decomposed_coordinates_and_features(self):
        """per-batch (coords[:, 1:], feats) lists (sunrgbd/brnetcanon.py:227)"""
        C, Fm = self.C, self.F
        nb = int(C[:, 0].max().item()) + 1
        cs, fs = [], []
        for b in range(nb):
            m = C[:, 0] == b
            cs.append(C[m][:, 1:])
            fs.append(Fm[m])
        return cs, fs

    def _like(self, F, tensor_stride=None, F_hl=None):
        t = SparseTensor(F, coordinate_manager=self.coordinate_manager,
                         tensor_stride=self.tensor_stride if tensor_stride is None else tensor_stride)
        t.F_hl = F_hl
        return t


def _workspace(dev, nbytes):
    """one persistent split-K workspace per (device, stream), grown on demand; stream-ordered reuse is safe
    because every conv of a stream finishes reading it before the next one starts."""
    return _lib.scratch(dev, "conv_ws", nbytes)


_range_flags = {}
_range_lock = threading.Lock()


def range_flag(dev):
    """one int32 word of pinned (device-visible) host memory per (device, stream): fp16-pair convolutions set it when
    an input magnitude leaves the fp16 range (cv_conv_desc.range_flag); read it after synchronising the stream"""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    with _range_lock:
        f = _range_flags.get(key)
        if f is None:
            f = _range_flags[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return f


def conv_forward(x_feats, weight, nbr, n_out, scale=None, shift=None, residual=None, relu=False,
                 out=None, flavour=0, row_perm=None, j_begin=0, j_end=0, acc_in=None, perm_groups=0,
                 cache_weights=True, pieces=None, in_hl=False, out_hl=False, res_hl=False, stem_mfma=False,
                 split_tickets=None, weight_t=False, acc_scale_dev=None):
    """Low-level call into cv_sp_conv_f32 (fused epilogue).  weight [K,Cin,Cout] or [Cin,Cout].
    in_hl / out_hl / res_hl: the operand is in the hl format (to_hl / from_hl; cv_conv_desc.in_hl), pieces=2 only.
    pieces=2: fp16-pair products (weights packed per call; the caller reads range_flag(dev) after synchronising);
    pieces=1: single bf16 products (the opt-in bf16 compute mode); None: 1 under COMPUTE_DTYPE == "bf16", else 3."""
    if pieces is None:
        pieces = 1 if COMPUTE_DTYPE == "bf16" else 3
    L = _lib.lib()
    dev = x_feats.device
    w = weight if weight.dim() == 3 else weight[None]
    if weight_t:
        # the TRANSPOSED convolution of `weight` ([K, Cout', Cin'] = a layer's forward kernel; the input gradient):
        # packed straight from the forward layout where the 16-bit piece path runs, transposed by a copy otherwise
        K, cout, cin = w.shape
        x6_ok = CONV_X6 and flavour in (0, 1) and cin % 32 == 0 and cout % 4 == 0 and pieces in (1, 2, 3)
        if not x6_ok:
            w = w.permute(0, 2, 1)
            weight_t = False
    else:
        K, cin, cout = w.shape
    assert x_feats.shape[1] == cin and x_feats.stride(1) == 1
    w = w.contiguous()
    if out is None:
        out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    if (AUTO_MASK_GROUPS > 1 and row_perm is None and perm_groups == 0 and flavour == 0 and nbr is not None
            and K == 27 and n_out >= AUTO_MASK_MIN_ROWS and cin % 32 == 0 and j_begin == 0 and j_end == 0
            and acc_in is None):
        # generic (module-by-module / training) path: mask-sorted offset groups, orders cached on the map
        row_perm = map_mask_perms(nbr, AUTO_MASK_GROUPS)
        perm_groups = row_perm.shape[0]
    wp6 = None
    acc_scale, flag = 0.0, None
    if CONV_X6 and flavour in (0, 1) and cin % 32 == 0 and cout % 4 == 0:
        if pieces == 2 and weight_t:
            # the input gradient on fp16 pairs: W^T packed straight from the forward layout with the layer's scale of this step
            pk = _bwd_ctx.get("packed")
            pre = pk.get((w.data_ptr(), 1)) if pk else None
            if pre is not None and pre[0].numel() == 2 * w.numel():
                wp6, k = pre                              # packed at the start of the step (_prepack_pairs)
                TRAIN_COUNTERS["prepacked"] += 1
            else:
                hints = _bwd_ctx.get("pair_scales")         # (the backward nodes run on the autograd engine's thread)
                k = hints.get(weight.data_ptr()) if hints is not None else None
                if k is None:
                    k = h2_scale_log2((w, None))
                wp6 = torch.empty(2 * w.numel(), dtype=torch.int16, device=dev)
                with _on(dev):
                    _lib.check(L.cv_sp_pack_weights_t_f32(_ptr(w), K, cout, cin, 2, int(k), _ptr(wp6), _stream(dev)),
                               "cv_sp_pack_weights_t_f32")
            acc_scale, flag = 2.0 ** -k, range_flag(dev)
        elif pieces == 2:
            key = id(weight)
            ver = (weight.data_ptr(), weight._version, tuple(weight.shape))
            hit = _packed_h2.get(key) if cache_weights else None
            pk = _bwd_ctx.get("packed") if not cache_weights else None
            pre = pk.get((w.data_ptr(), 0)) if pk else None
            if pre is not None and pre[0].numel() == 2 * w.numel():
                hit = (ver, pre[0], pre[1])               # packed at the start of the step (_prepack_pairs)
                TRAIN_COUNTERS["prepacked"] += 1
            elif hit is None or hit[0] != ver:
                hints = getattr(_train_state, "pair_scales", None)
                k = hints.get(weight.data_ptr()) if hints is not None else None
                if k is None:
                    k = h2_scale_log2((w, None))          # (one host wait: the training step hands the scales over instead)
                hit = (ver, packed_weights_h2(w, None, k), k)
                if cache_weights:
                    if len(_packed_h2) > 512:
                        _packed_h2.clear()
                    _packed_h2[key] = hit
            wp6, acc_scale, flag = hit[1], 2.0 ** -hit[2], range_flag(dev)
        elif weight_t:
            wp6 = torch.empty((3 if pieces == 3 else 1) * w.numel(), dtype=torch.int16, device=dev)
            with _on(dev):
                _lib.check(L.cv_sp_pack_weights_t_f32(_ptr(w), K, cout, cin, pieces, 0, _ptr(wp6), _stream(dev)),
                           "cv_sp_pack_weights_t_f32")
        elif pieces == 1:
            wp6 = packed_weights_bf16(w)
        else:
            wp6 = packed_weights_x6(weight, w, cache_weights)
    if stem_mfma and pieces == 2 and cin in (3, 6) and cout == 32 and K <= 128 and nbr is not None and flavour == 0:
        # the matrix-core stem (conv_stem_mfma): BatchNorm scale folded into the fp16-pair weights
        hints = getattr(_train_state, "pair_scales", None) if scale is None else None      # (a training step hands the scales over)
        k = hints.get(weight.data_ptr()) if hints is not None else None
        if k is None:
            k = h2_scale_log2((w, scale))
        wp6, acc_scale, flag, scale = packed_weights_stem_h2(w, scale, k), 2.0 ** -k, range_flag(dev), None
    ws = None
    if perm_groups > 1:
        ws = _workspace(dev, 4 * perm_groups * n_out * cout + 256)
    elif flavour == 0 and n_out < 128 * 384:
        ws = _workspace(dev, int(L.cv_sp_conv_workspace_bytes(n_out, cout, K)))
    p = lambda t: t.data_ptr() if t is not None else None
    d = _lib.ConvDesc(p(x_feats), x_feats.shape[0], x_feats.stride(0), cin, p(w), K, cout, p(nbr), n_out,
                      p(scale), p(shift), p(residual), residual.stride(0) if residual is not None else 0,
                      1 if relu else 0, p(out), out.stride(0), flavour, p(ws),
                      ws.numel() if ws is not None else 0, p(row_perm), j_begin, j_end, p(acc_in),
                      acc_in.stride(0) if acc_in is not None else 0, perm_groups,
                      None, None, None, p(wp6), None, 0, 0, None,
                      1 if (perm_groups > 1 and getattr(row_perm, "_cv_has_map", False)) else 0,
                      2 if flag is not None else (1 if (pieces == 1 and wp6 is not None) else 0), acc_scale, p(flag),
                      1 if in_hl else 0, 1 if out_hl else 0, 1 if res_hl else 0, p(split_tickets), p(acc_scale_dev))
    with _on(dev):
        _lib.check(L.cv_sp_conv_f32(ctypes.byref(d), _stream(dev)), "cv_sp_conv_f32")
    return out


_policy_tls = threading.local()


def masked_min_rows():
    """Rows from which a level's 3x3x3 convolutions run mask-sorted: the CALLING THREAD's value while it is inside
    pipeline.scene_policy(...) (launch sizing per call: two hosts with different policies in one process do not see each
    other), else the process-wide CoordinateManager.MASKED_MIN_ROWS."""
    v = getattr(_policy_tls, "masked_min_rows", None)
    return CoordinateManager.MASKED_MIN_ROWS if v is None else v


def set_masked_min_rows_thread(rows):
    """the calling thread's masked_min_rows() (None: follow the process-wide value); returns the previous thread value"""
    before = getattr(_policy_tls, "masked_min_rows", None)
    _policy_tls.masked_min_rows = None if rows is None else int(rows)
    return before


def set_split_target(workgroups):
    """cv_sp_set_split_target: workgroups a split convolution launch aims at (0: the library default, 768 - best for one
    scene in flight; 256 pays from about four scenes in flight).  Returns the previous value."""
    return int(_lib.lib().cv_sp_set_split_target(int(workgroups)))


def set_option(name, value):
    """cv_sp_set_option: kernel selection knobs of the convolutions ("hd_mask", "hd_min_rows"); returns the previous value"""
    import ctypes
    prev = ctypes.c_longlong(0)
    _lib.check(_lib.lib().cv_sp_set_option(name.encode(), int(value), ctypes.byref(prev)), "cv_sp_set_option")
    return int(prev.value)


def option(name):
    """current value of a cv_sp_set_option knob"""
    v = ctypes.c_longlong(0)
    _lib.check(_lib.lib().cv_sp_get_option(name.encode(), ctypes.byref(v)), "cv_sp_get_option")
    return int(v.value)


def to_hl(x):
    """fp32 rows [n, C] (C % 32 == 0) -> the hl format (same shape and dtype, the bytes hold the fp16 pairs)"""
    y = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.lib().cv_sp_to_hl_f32(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), y.data_ptr(), y.stride(0),
                                              range_flag(x.device).data_ptr(), _stream(x.device)), "cv_sp_to_hl_f32")
    return y


def from_hl(x):
    y = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.check(_lib.lib().cv_sp_from_hl_f32(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), y.data_ptr(),
                                                y.stride(0), _stream(x.device)), "cv_sp_from_hl_f32")
    return y


# conv_forward applies mask-sorted offset groups to big 3x3x3 maps on its own (0 = off)
AUTO_MASK_GROUPS = int(os.environ.get("CV_MASK_GROUPS", "4"))
AUTO_MASK_MIN_ROWS = int(os.environ.get("CV_AUTO_MASK_MIN_ROWS", "16384"))


def map_mask_perms(nbr, groups):
    """[groups, n] processing orders of a kernel map (see CoordinateManager.mask_perms), cached on the map."""
    hit = getattr(nbr, "_cv_mask_perms", None)
    if hit is not None and getattr(hit, "_cv_from_plan", False):
        return hit          # built with the scene's coordinate plan (its own group count): no second sort of the same map
    if hit is None or hit.shape[0] != groups:
        hit = nbr._cv_mask_perms = _perms_with_map(nbr, groups)
    return hit


def _perms_with_map(nbr, groups):
    """[groups, n] orders (a view) followed, in the same buffer, by the kernel map rows of every group in processing
    order [groups, n, W] (cv_sp_mask_perms with_map = 1); the view carries `_cv_has_map` for conv_forward."""
    L = _lib.lib()
    n, K = nbr.shape
    W = (K + groups - 1) // groups
    # orders, the map rows in processing order, then one validity byte per (group, row) (cv_sp_mask_perms with_map = 1)
    flat = torch.empty(groups * n * (1 + W) + (groups * n + 3) // 4, dtype=torch.int32, device=nbr.device)
    ws = torch.empty(groups * 4096, dtype=torch.uint8, device=nbr.device)
    with _on(nbr.device):
        _lib.check(L.cv_sp_mask_perms(_ptr(nbr), n, K, groups, _ptr(flat), _ptr(ws), ws.numel(), 1,
                                      _stream(nbr.device)), "cv_sp_mask_perms")
    m = flat[:groups * n].view(groups, n)
    m._cv_has_map = True
    m._cv_flat = flat
    return m


_packed = {}
_packed_x6 = {}
_packed_h2 = {}
# 1 (default): vector-path convs run their fp32 products as six bf16 piece products on the bf16 matrix cores
# (conv_rows_wp); 0: v_mfma_f32_32x32x2_f32
CONV_X6 = os.environ.get("CV_CONV_X6", "1") != "0"


# Product precision of the vector-path convolutions (forward, input gradient, weight gradient):
#   "fp32" (default): fp32-level products (bf16 triples / fp16 pairs, see sparse_conv.hip) - the parity path;
#   "bf16": operands rounded to bf16, one bf16 x bf16 product, fp32 accumulation and fp32 storage - the opt-in
#           compute mode BASELINE configs 3-4 name for training (bench.py --dtype bf16); NOT within the 1e-4 bar.
COMPUTE_DTYPE = "bf16" if os.environ.get("CV_COMPUTE_DTYPE", "fp32") == "bf16" else "fp32"


def set_compute_dtype(dtype):
    """"fp32" or "bf16" (see COMPUTE_DTYPE); returns the previous setting.  Cached eval programs are keyed by it."""
    global COMPUTE_DTYPE
    assert dtype in ("fp32", "bf16"), dtype
    prev, COMPUTE_DTYPE = COMPUTE_DTYPE, dtype
    return prev


def packed_weights_bf16(w3, col_scale=None):
    """uncached: weights (times an optional per-output-column scale) rounded to bf16 in conv_rows_wp's layout"""
    L = _lib.lib()
    K, cin, cout = w3.shape
    wp = torch.empty(w3.numel(), dtype=torch.int16, device=w3.device)
    with _on(w3.device):
        _lib.check(L.cv_sp_pack_weights_bf16_f32(_ptr(w3), K, cin, cout, _ptr(col_scale), _ptr(wp),
                                                 _stream(w3.device)), "cv_sp_pack_weights_bf16_f32")
    return wp


def invalidate_weight_caches():
    """Drop every cached re-packing of a weight tensor.  The caches are keyed by tensor version, and fused
    optimizers (torch.optim.Adam(fused=True)) update parameters WITHOUT bumping it: modules call this when they
    switch between train and eval mode, and the training path never caches."""
    _packed.clear()
    _packed_x6.clear()
    _packed_h2.clear()


def packed_weights_x6_scaled(w3, col_scale):
    """uncached: weights times a per-output-column scale (a folded BatchNorm), split into bf16 pieces"""
    L = _lib.lib()
    K, cin, cout = w3.shape
    wp = torch.empty(3 * w3.numel(), dtype=torch.int16, device=w3.device)
    with _on(w3.device):
        _lib.check(L.cv_sp_pack_weights_x6_f32(_ptr(w3), K, cin, cout, _ptr(col_scale), _ptr(wp), _stream(w3.device)),
                   "cv_sp_pack_weights_x6_f32")
    return wp


def h2_scale_log2(*scaled_weights):
    """power of two that brings the largest magnitude of the given (weight [K,Cin,Cout], column scale or None) sets
    to about 2^13 (cv_sp_pack_weights_h2_f32); one host sync, at pack time only"""
    import math
    amax = 0.0
    for w3, col_scale in scaled_weights:
        a = w3.abs().amax(dim=(0, 1)) if col_scale is not None else w3.abs().max()
        amax = max(amax, float((a * col_scale.abs()).max()) if col_scale is not None else float(a))
    if not math.isfinite(amax) or amax <= 0.0:
        return 0
    return max(-60, min(60, 13 - math.ceil(math.log2(amax))))


def packed_weights_h2(w3, col_scale, scale_log2):
    """uncached: weights (times an optional per-output-column scale and 2^scale_log2) split into two fp16 pieces"""
    L = _lib.lib()
    K, cin, cout = w3.shape
    wp = torch.empty(2 * w3.numel(), dtype=torch.int16, device=w3.device)
    with _on(w3.device):
        _lib.check(L.cv_sp_pack_weights_h2_f32(_ptr(w3), K, cin, cout, _ptr(col_scale), int(scale_log2), _ptr(wp),
                                               _stream(w3.device)), "cv_sp_pack_weights_h2_f32")
    return wp


def packed_weights_stem_h2(w3, col_scale, scale_log2):
    """uncached: stem weights [K, 3|6, 32] (times an optional column scale and 2^scale_log2) as fp16 pairs in the
    B-operand order of conv_stem_mfma (cv_sp_pack_weights_stem_h2_f32)"""
    L = _lib.lib()
    K, cin, cout = w3.shape
    assert cout == 32
    wp = torch.empty(8 * cin * 1024, dtype=torch.int16, device=w3.device)
    with _on(w3.device):
        _lib.check(L.cv_sp_pack_weights_stem_h2_f32(_ptr(w3), K, cin, _ptr(col_scale), int(scale_log2), _ptr(wp),
                                                    _stream(w3.device)), "cv_sp_pack_weights_stem_h2_f32")
    return wp


def packed_weights_x6(weight, w3, cache=True):
    """weights split into bf16 pieces for conv_rows_wp (cv_sp_pack_weights_x6_f32), cached per parameter tensor and
    re-packed when it is modified in place or re-allocated."""
    key = id(weight)
    ver = (weight.data_ptr(), weight._version, tuple(weight.shape))
    hit = _packed_x6.get(key)
    if not cache:            # training: the weights change every step (see invalidate_weight_caches)
        hit = None
    if hit is None or hit[0] != ver or hit[2]() is not weight:
        import weakref
        L = _lib.lib()
        K, cin, cout = w3.shape
        wp = torch.empty(3 * w3.numel(), dtype=torch.int16, device=w3.device)
        with _on(w3.device):
            _lib.check(L.cv_sp_pack_weights_x6_f32(_ptr(w3), K, cin, cout, None, _ptr(wp), _stream(w3.device)),
                       "cv_sp_pack_weights_x6_f32")
        try:
            ref = weakref.ref(weight, lambda _r, k=key: _packed_x6.pop(k, None))
        except TypeError:
            ref = lambda: weight
        hit = _packed_x6[key] = (ver, wp, ref)
    return hit[1]



def conv_forward_masked(x_feats, weight, nbr, perms, n_out, **epilogue):
    """k^3 conv with the kernel offsets split into G groups, each group processed with the rows in the
    order sorted by that group's neighbour mask (see sparse_conv.hip); one launch + one reduce.
    perms: int32 [G, n_out] from CoordinateManager.mask_perms."""
    return conv_forward(x_feats, weight, nbr, n_out, row_perm=perms, perm_groups=perms.shape[0],
                        flavour=epilogue.pop("flavour", 0), **epilogue)


def transposed_map(nbr, n_in):
    """nbr_t [n_in, K] with nbr_t[i][j] = u iff nbr[u][j] == i (cached on the map tensor)."""
    hit = getattr(nbr, "_cv_transposed", None)
    if hit is None or hit.shape[0] != n_in:
        L = _lib.lib()
        hit = torch.empty((n_in, nbr.shape[1]), dtype=torch.int32, device=nbr.device)
        with _on(nbr.device):
            _lib.check(L.cv_sp_transpose_map(_ptr(nbr), nbr.shape[0], nbr.shape[1], n_in, _ptr(hit),
                                             _stream(nbr.device)), "cv_sp_transpose_map")
        nbr._cv_transposed = hit
    return hit


def conv_wgrad(x_feats, grad_out, nbr, K):
    """dW [K, Cin, Cout] = sum_u x[nbr[u][j]]^T dy[u]  (cv_sp_conv_wgrad_f32)."""
    L = _lib.lib()
    dev = x_feats.device
    cin, cout, n_out = x_feats.shape[1], grad_out.shape[1], grad_out.shape[0]
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    ws = _workspace(dev, int(L.cv_sp_wgrad_workspace_bytes(n_out, cin, cout, K)))
    with _on(dev):
        if COMPUTE_DTYPE == "bf16":
            _lib.check(L.cv_sp_conv_wgrad_px_f32(_ptr(x_feats), x_feats.stride(0), cin, _ptr(grad_out),
                                                 grad_out.stride(0), cout, _ptr(nbr), K, n_out, _ptr(dw), _ptr(ws),
                                                 ws.numel(), 1, _stream(dev)), "cv_sp_conv_wgrad_px_f32")
        else:
            _lib.check(L.cv_sp_conv_wgrad_f32(_ptr(x_feats), x_feats.stride(0), cin, _ptr(grad_out),
                                              grad_out.stride(0), cout, _ptr(nbr), K, n_out, _ptr(dw), _ptr(ws),
                                              ws.numel(), _stream(dev)), "cv_sp_conv_wgrad_f32")
    return dw


def col_sum(x):
    """column sums (bias gradient) in a fixed summation order: cv_sp_col_sum_det_f32"""
    L = _lib.lib()
    out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    ws = _lib.scratch(x.device, "col_sum", int(L.cv_sp_col_sum_workspace_bytes(x.shape[0], x.shape[1])))
    with _on(x.device):
        _lib.check(L.cv_sp_col_sum_det_f32(_ptr(x), x.shape[0], x.shape[1], x.stride(0), _ptr(out), _ptr(ws), ws.numel(),
                                           _stream(x.device)), "cv_sp_col_sum_det_f32")
    return out


# training: weight gradient of a layer on a side stream next to its input gradient.
#   CV_BACKWARD_OVERLAP=0: one stream; 1: the layer's stream waits for the side stream before backward() returns;
#   2 (default): that wait moves to the end of the backward pass (the side stream works through the weight gradients
#   while the layer's stream goes on with the BatchNorm backward and the next input gradient) whenever nothing can
#   touch the gradient earlier - see _ConvFn.backward.
BACKWARD_OVERLAP = int(os.environ.get("CV_BACKWARD_OVERLAP", "2"))
LATE_GRAD_LOG = None        # tests: a list that receives (kernel parameter, data_ptr of d_kernel) of every late-joined layer
_wgrad_streams = {}
_wgrad_lock = threading.Lock()
_wgrad_pending = threading.local()


def _wgrad_stream(dev):
    """the side stream of (device, current stream) for _ConvFn.backward"""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    with _wgrad_lock:
        s = _wgrad_streams.get(key)
        if s is None:
            s = _wgrad_streams[key] = torch.cuda.Stream(device=dev)
    return s


def _join_at_end_of_backward(cur, side):
    """queue ONE callback per backward pass and (stream, side stream) pair: the stream waits for the side stream when
    the autograd engine has run the last node (before it hands the gradients to the caller's stream)"""
    pend = getattr(_wgrad_pending, "pairs", None)
    if pend is None:
        pend = _wgrad_pending.pairs = {}
    # (the pass is part of the key: two backward passes interleaved on one engine thread each get their own join)
    key = (torch._C._current_graph_task_id(), cur.cuda_stream, side.cuda_stream)
    if key in pend:
        return
    pend[key] = (cur, side)

    def join():
        pend.pop(key, None)
        cur.wait_stream(side)
        torch.cuda.current_stream(cur.device).wait_stream(side)

    torch.autograd.Variable._execution_engine.queue_callback(join)


def _gradient_untouched_until_end(kernel):
    """True when nothing reads or writes the weight gradient between _ConvFn.backward and the end of the backward pass:
    no gradient to accumulate into (AccumulateGrad then keeps the tensor itself, no kernel), no hooks on the parameter,
    no process group with more than one rank (DDP's reducer copies gradients into its buckets as they arrive), no
    anomaly / graph-building mode.  Code that hooks the gradient accumulators by other means sets
    CV_BACKWARD_OVERLAP=1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return False          # (DDP's reducer hooks sit on the gradient accumulators, invisible from here - also with ONE rank,
                              #  where the bucket copy still runs on the layer's stream as the gradient arrives)
    # (AccumulateGrad keeps a gradient tensor as it is - no kernel on the layer's stream - only when its layout is the
    # parameter's: a non-contiguous kernel parameter would make it clone d_kernel on the main stream, racing the side stream)
    return (kernel.is_leaf and kernel.grad is None and kernel.is_contiguous() and not kernel._backward_hooks
            and not getattr(kernel, "_post_accumulate_grad_hooks", None)
            and not torch.is_grad_enabled() and not torch.is_anomaly_enabled())


TRAIN_FWD_PIECES = int(os.environ.get("CV_TRAIN_FWD_PIECES", "3"))      # 3 (default): bf16 triples; 2: fp16 pairs in the training forward (measured: no gain, profiles/r4/train_ab.txt)
# 1 (default): the training forward's convolutions run on the eval path's hl-format kernels (conv_hl / conv_hd: fp16 pairs, operands
# split once by the producing BatchNorm pass instead of once per gather) wherever the input has an hl twin; the range flag guards
# it like TRAIN_FWD_PIECES = 2 (train.train_step redoes the step on the bf16 triples).  0: conv_rows_wp on the fp32 rows.
TRAIN_FWD_HL = int(os.environ.get("CV_TRAIN_FWD_HL", "1"))
_train_state = threading.local()


def train_forward_hl():
    return TRAIN_FWD_HL != 0 and COMPUTE_DTYPE != "bf16"


# 1 (default): the input gradients run on the hl-format kernels too.  The BatchNorm backward writes dx a second time as fp16 pairs
# times a per-layer power of two taken from the PREVIOUS step's largest |dx| of that layer (three words per layer that live on
# the device: cv_sp_bn_backward_hl_f32) - inside train.train_step only (ME.pair_scale_hints rotates the maxima between steps);
# a layer's first step, and every backward outside a step, takes the bf16 triples.
TRAIN_BWD_HL = int(os.environ.get("CV_TRAIN_BWD_HL", "1"))


# 1 (default): train.train_step packs the fp16-pair weights of all convolutions (both directions) by one launch per step
TRAIN_PREPACK = int(os.environ.get("CV_TRAIN_PREPACK", "1"))
TRAIN_COUNTERS = {"hl_dgrad": 0, "prepacked": 0}           # (tests: how many input gradients took the hl path)
# what the backward nodes of the step in progress share.  Process-wide, not thread-local: the autograd engine runs the nodes on
# its own thread.  One training step at a time per process (the one-process-per-GPU layout of train_joint.py).
_bwd_ctx = {"slots": None, "twins": {}, "pair_scales": None, "packed": None}


def _prepack_pairs(module, ks, scales, dev):
    """fp16-pair weights of every convolution of a training step - forward layout and, for the input gradients, transposed -
    packed by ONE launch in front of the step's forward (cv_sp_pack_weights_h2_batch_f32) into an arena that lives with the
    module: {(kernel.data_ptr(), transposed): (packed words, scale_log2)} for conv_forward.  Per layer and direction this
    was a launch of its own (117 per MinkUNet34C step) with the stream idle in front of most of them."""
    L = _lib.lib()
    sig = tuple((k.data_ptr(), tuple(k.shape), k.is_contiguous()) for k in ks) + (bool(TRAIN_BWD_HL),)
    st = module.__dict__.get("_prepack_state")
    if st is None or st["sig"] != sig or st["arena"].device != dev:
        meta, off = [], 0
        for k in ks:
            if not k.is_contiguous():
                continue
            K, cin, cout = (k.shape if k.dim() == 3 else (1,) + tuple(k.shape))
            words = 2 * K * cin * cout
            if cin % 32 == 0 and cout % 4 == 0:
                meta.append((k.data_ptr(), 0, K, cin, cout, off, words))
                off += (words + 63) // 64 * 64
            if TRAIN_BWD_HL and cout % 32 == 0 and cin % 4 == 0:      # the transposed convolution: Cin' = cout, Cout' = cin
                meta.append((k.data_ptr(), 1, K, cout, cin, off, words))
                off += (words + 63) // 64 * 64
        arena = torch.empty(max(off, 64), dtype=torch.int16, device=dev)
        jobs = (_lib.PackJob * max(len(meta), 1))()
        for j, (ptr, trans, K, cin, cout, o, words) in zip(jobs, meta):
            j.w, j.wp, j.K, j.cin, j.cout, j.trans = ptr, arena.data_ptr() + 2 * o, K, cin, cout, trans
        st = module.__dict__["_prepack_state"] = {
            "sig": sig, "meta": meta, "arena": arena, "jobs": jobs,
            "d_jobs": torch.empty(max(len(meta), 1) * ctypes.sizeof(_lib.PackJob), dtype=torch.uint8, device=dev)}
    meta, arena, jobs = st["meta"], st["arena"], st["jobs"]
    packed = {}
    for j, (ptr, trans, K, cin, cout, o, words) in zip(jobs, meta):
        j.scale_log2 = int(scales.get(ptr, 0))
        packed[(ptr, trans)] = (arena[o:o + words], j.scale_log2)
    if meta:
        with _on(dev):
            _lib.check(L.cv_sp_pack_weights_h2_batch_f32(jobs, len(meta), ctypes.c_void_p(st["d_jobs"].data_ptr()), _stream(dev)),
                       "cv_sp_pack_weights_h2_batch_f32")
    return packed


class _GradSlots:
    """per-layer words of the hl-format gradient twins (keyed by the BatchNorm gain's storage), one device buffer per module tree"""

    def __init__(self, dev):
        self.buf = torch.zeros((128, 8200), dtype=torch.int32, device=dev)  # [layer][4096 x this step's maxima | 4096 x last step's | 1 / s] (CV_BN_SLOT_WORDS)
        self.index = {}
        self.seen = set()                    # layers whose slot[1] holds a maximum (they ran a backward in the step before)
        self.running = set()

    def rotate(self):
        with torch.no_grad():
            self.buf[:, 4096:8192].copy_(self.buf[:, 0:4096])
            self.buf[:, 0:4096].zero_()
        self.seen, self.running = self.running, set()

    def take(self, key):
        """(slot tensor, usable) for the layer; usable = last step left a maximum"""
        i = self.index.get(key)
        if i is None:
            if len(self.index) >= self.buf.shape[0]:
                return None, False
            i = self.index[key] = len(self.index)
        self.running.add(key)
        return self.buf[i], key in self.seen


def train_uses_pairs():
    """the training forward multiplies fp16 pairs somewhere (range flag to check, weight scales to hand over)"""
    return COMPUTE_DTYPE != "bf16" and (TRAIN_FWD_PIECES == 2 or TRAIN_FWD_HL != 0)


class pair_scale_hints:
    """power-of-two weight scales of the fp16-pair products for every convolution of `module`, valid inside the `with` block (a
    training step: the weights do not move between its forward and its optimizer step).  Without it every fp16-pair convolution
    whose packed weights are not cached reads its own maximum back (63 host waits per MinkUNet34C forward).
    No host wait in the steady state: the max-norms (ONE multi-tensor launch) travel to pinned memory behind the step that
    computed them and the NEXT step reads them - an optimizer step moves a weight maximum by a fraction of a percent and the scale
    leaves a factor of 8 to the fp16 range.  The first step of a module (or a changed parameter list) waits once."""

    def __init__(self, module):
        self.module = module

    def __enter__(self):
        self.outer = getattr(_train_state, "pair_scales", None)
        ks = [m.kernel for m in self.module.modules() if isinstance(m, MinkowskiConvolutionBase) and m.kernel.is_cuda]
        scales = {}
        if ks and train_uses_pairs():
            ptrs = tuple(k.data_ptr() for k in ks)
            dev = ks[0].device
            st = self.module.__dict__.get("_pair_scale_state")
            if st is None or st["ptrs"] != ptrs:
                st = self.module.__dict__["_pair_scale_state"] = {
                    "ptrs": ptrs, "host": torch.empty(len(ks), dtype=torch.float32).pin_memory(), "event": None}
                self._request(st, ks, dev)
            st["event"].synchronize()                # (already passed in the steady state: it was recorded a step ago)
            amax = st["host"].tolist()
            self._request(st, ks, dev)               # for the next step
            for k, a in zip(ks, amax):
                scales[k.data_ptr()] = (max(-60, min(60, 13 - math.ceil(math.log2(a)))) if (math.isfinite(a) and a > 0) else 0)
        _train_state.pair_scales = scales
        self.outer_slots = _bwd_ctx["slots"]
        _bwd_ctx["slots"], _bwd_ctx["twins"] = None, {}
        self.outer_packed = _bwd_ctx["packed"]
        _bwd_ctx["packed"] = _prepack_pairs(self.module, ks, scales, ks[0].device) if (ks and train_forward_hl() and TRAIN_PREPACK) else None
        if ks and train_forward_hl() and TRAIN_BWD_HL:
            slots = self.module.__dict__.get("_grad_slots")
            if slots is None or slots.buf.device != ks[0].device:
                slots = self.module.__dict__["_grad_slots"] = _GradSlots(ks[0].device)
            slots.rotate()
            _bwd_ctx["slots"] = slots
            _bwd_ctx["pair_scales"] = scales
        return self

    @staticmethod
    def _request(st, ks, dev):
        with torch.no_grad():
            amax = torch.stack(torch._foreach_norm([k.detach() for k in ks], float("inf")))
            st["host"].copy_(amax, non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record(torch.cuda.current_stream(dev))

    def __exit__(self, *exc):
        _train_state.pair_scales = self.outer
        _bwd_ctx["slots"], _bwd_ctx["twins"], _bwd_ctx["pair_scales"] = self.outer_slots, {}, None
        _bwd_ctx["packed"] = self.outer_packed
        return False


def training_range_flag_device(dev):
    """the range flag of this thread's stream as a float32 device scalar (1.0 = a training forward since the last reset left the
    fp16 range), read in stream order behind the kernels that may raise it: hand it to a fused optimizer as `found_inf` and the
    update is skipped on the device without a host wait (train.train_step).  None when no forward used the fp16 pairs."""
    if not getattr(_train_state, "used_pairs", False):
        return None
    return range_flag(dev).to(dev, non_blocking=True).to(torch.float32).reshape(())


def training_range_flag_peek(dev):
    """the flag's current value without waiting for anything (kernels still in flight may raise it later)"""
    return int(range_flag(dev)[0]) != 0


def training_forward_left_fp16_range(dev, group=None):
    """True when a convolution of this thread's training forwards since the last call staged an input beyond the fp16
    range on the current stream (synchronises the stream; the flag is reset).  False without a wait when no forward used
    the fp16 pairs.  group: a process group whose ranks step together (DDP) - the answer is then the MAXIMUM over the ranks
    (what GradScaler does with found_inf): every rank redoes the step, or none does."""
    used = getattr(_train_state, "used_pairs", False)
    if not used and group is None:
        return False
    _train_state.used_pairs = False
    torch.cuda.current_stream(dev).synchronize()
    flag = range_flag(dev)
    raised = int(flag[0]) != 0
    if group is not None:
        import torch.distributed as dist
        t = torch.tensor([1.0 if raised else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        raised = float(t[0]) != 0.0
    if not raised:
        return False
    flag.zero_()
    return True


def copy_unless_flag(srcs, dsts, dev, flag=None):
    """dsts[i].copy_(srcs[i]) for lists of same-shaped device tensors in ceil(n / 96) launches, skipped ON THE DEVICE while
    `flag` (a device-visible int32 tensor, e.g. range_flag(dev); None: always copy) is non-zero: cv_sp_copy_unless_flag."""
    n = len(srcs)
    if n == 0:
        return
    vp = ctypes.c_void_p
    a_src = (vp * n)(*[t.data_ptr() for t in srcs])
    a_dst = (vp * n)(*[t.data_ptr() for t in dsts])
    a_len = (ctypes.c_longlong * n)(*[t.numel() * t.element_size() for t in srcs])
    with _on(dev):
        _lib.check(_lib.lib().cv_sp_copy_unless_flag(a_src, a_dst, a_len, n, vp(flag.data_ptr()) if flag is not None else None,
                                                     _stream(dev)), "cv_sp_copy_unless_flag")


class _ConvFn(torch.autograd.Function):
    """Sparse convolution with HIP forward, input-gradient (the same kernel on the transposed map with
    transposed weights) and weight-gradient kernels."""

    @staticmethod
    def forward(ctx, feats, kernel, bias, nbr, n_out, feats_hl=None):
        feats = feats.contiguous()
        ctx.save_for_backward(feats, kernel, nbr if nbr is not None else torch.empty(0))
        ctx.has_nbr = nbr is not None
        ctx.has_bias = bias is not None
        shift = bias.reshape(-1).contiguous() if bias is not None else None
        # fp32-level products of the training FORWARD: fp16 pairs (three piece products) while the activations are inside
        # the fp16 range - they sit behind BatchNorm - with the range flag as the guard (train.train_step reads it once per
        # step and redoes the step on the bf16 triples); the input gradient keeps the triples (gradients have no bound)
        if feats_hl is not None and train_forward_hl() and feats.shape[1] % 32 == 0 and feats_hl.stride(0) % 32 == 0:
            # the eval path's kernels on the hl twin (the fp32 rows stay saved for the weight gradient)
            _train_state.used_pairs = True
            return conv_forward(feats_hl, kernel, nbr, n_out, shift=shift, cache_weights=False, pieces=2, in_hl=True)
        k3 = kernel if kernel.dim() == 3 else kernel[None]
        if (train_forward_hl() and nbr is not None and feats.shape[1] in (3, 6) and k3.shape[2] == 32 and k3.shape[0] <= 128):
            # the stem (conv0p1s1: 5x5x5, 3 or 6 -> 32 channels) on the matrix cores, like the eval program's (conv_stem_mfma)
            _train_state.used_pairs = True
            return conv_forward(feats, kernel, nbr, n_out, shift=shift, cache_weights=False, pieces=2, stem_mfma=True)
        pieces = None
        if TRAIN_FWD_PIECES == 2 and COMPUTE_DTYPE != "bf16" and feats.shape[1] % 32 == 0:
            pieces = 2
            _train_state.used_pairs = True
        return conv_forward(feats, kernel, nbr, n_out, shift=shift, cache_weights=False, pieces=pieces)

    @staticmethod
    def backward(ctx, grad):
        feats, kernel, nbr = ctx.saved_tensors
        nbr = nbr if ctx.has_nbr else None
        grad = grad.contiguous()
        k3 = kernel if kernel.dim() == 3 else kernel[None]
        d_feats = d_kernel = d_bias = None
        # The two gradients of a layer read the same dy and write different tensors: the weight gradient goes to a
        # side stream next to the input gradient (each alone leaves most SIMDs with one or two waves and a tail of
        # long tasks).  Either the layer's own stream waits for the side stream before backward returns (everything
        # that consumes d_kernel, frees x / dy or reuses their memory is then ordered behind both), or - when nothing
        # can touch the gradient before the end of the pass - that wait is queued for the end of the backward pass and
        # x / dy are marked as in use by the side stream.  The same kernels on the same inputs: results bit for bit.
        side = _wgrad_stream(grad.device) if (BACKWARD_OVERLAP and ctx.needs_input_grad[0] and
                                              ctx.needs_input_grad[1]) else None
        late = False
        if side is not None:
            cur = torch.cuda.current_stream(grad.device)
            late = BACKWARD_OVERLAP >= 2 and _gradient_untouched_until_end(kernel)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                d_kernel = conv_wgrad(feats, grad, nbr, k3.shape[0]).reshape(kernel.shape)
            if late:
                # the side stream still reads x and dy after this node has released them: their blocks go back to the
                # allocator only when the side stream has passed this point
                feats.record_stream(side)
                grad.record_stream(side)
                if nbr is not None:
                    nbr.record_stream(side)
                _join_at_end_of_backward(cur, side)
                if LATE_GRAD_LOG is not None:
                    LATE_GRAD_LOG.append((kernel, d_kernel.data_ptr()))
        twins = _bwd_ctx["twins"]
        twin = twins.pop(grad.data_ptr(), None) if twins else None          # (taken out whether or not it is used)
        if ctx.needs_input_grad[0]:
            nbr_t = transposed_map(nbr, feats.shape[0]) if nbr is not None else None
            if (twin is not None and twin[2] == grad.shape and COMPUTE_DTYPE != "bf16" and k3.shape[2] % 32 == 0
                    and k3.shape[1] % 4 == 0):
                # the eval path's kernels on the gradient's hl twin (written by the BatchNorm backward above this layer)
                d_feats = conv_forward(twin[0], k3.detach(), nbr_t, feats.shape[0], cache_weights=False, weight_t=True,
                                       pieces=2, in_hl=True, acc_scale_dev=twin[1])
                TRAIN_COUNTERS["hl_dgrad"] += 1
            else:
                d_feats = conv_forward(grad, k3.detach(), nbr_t, feats.shape[0], cache_weights=False, weight_t=True)
        if side is not None:
            if not late:
                cur.wait_stream(side)
        elif ctx.needs_input_grad[1]:
            d_kernel = conv_wgrad(feats, grad, nbr, k3.shape[0]).reshape(kernel.shape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            d_bias = col_sum(grad).reshape(1, -1)
        return d_feats, d_kernel, d_bias, None, None, None


class MinkowskiConvolutionBase(nn.Module):
    TRANSPOSED = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, dimension=3, **_):
        super().__init__()
        assert dimension == 3, "only D=3 is built"
        assert dilation == 1, "dilation is not used by the reference network"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = kernel_size if isinstance(kernel_size, int) else int(kernel_size[0])
        self.stride = stride if isinstance(stride, int) else int(stride[0])
        self.dimension = dimension
        K = self.kernel_size ** 3
        shape = (in_channels, out_channels) if K == 1 else (K, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty((1, out_channels), dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        K = self.kernel_size ** 3
        n = (self.out_channels if self.TRANSPOSED else self.in_channels) * K
        stdv = 1.0 / math.sqrt(n)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def train(self, mode=True):
        invalidate_weight_caches()
        return super().train(mode)

    def _map(self, x):
        cm, ts = x.coordinate_manager, x.tensor_stride
        if self.TRANSPOSED:
            assert self.kernel_size == 2 and self.stride == 2, "only k2s2 transposed conv is built"
            return cm.up_map(ts), ts // 2
        if self.kernel_size == 1 and self.stride == 1:
            return None, ts
        return cm.kernel_map(self.kernel_size, ts, self.stride), ts * self.stride

    def forward(self, x):
        nbr, ts_out = self._map(x)
        n_out = x.coordinate_manager.num_rows(ts_out)
        F = _ConvFn.apply(x.F, self.kernel, self.bias, nbr, n_out, getattr(x, "F_hl", None))
        return x._like(F, ts_out)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d" % (self.in_channels, self.out_channels,
                                                             self.kernel_size, self.stride)


class MinkowskiConvolution(MinkowskiConvolutionBase):
    TRANSPOSED = False


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    TRANSPOSED = True


def bn_affine(bn):
    """(scale, shift) of an eval-mode BatchNorm1d, computed on the device."""
    L = _lib.lib()
    c = bn.num_features
    dev = bn.weight.device
    out = torch.empty((2, c), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(L.cv_sp_bn_fold_f32(_ptr(bn.weight), _ptr(bn.bias), _ptr(bn.running_mean),
                                       _ptr(bn.running_var), _ptr(None), float(bn.eps), c, _ptr(out[0]),
                                       _ptr(out[1]), _stream(dev)), "cv_sp_bn_fold_f32")
    return out[0], out[1]


def affine_forward(F, scale, shift, relu, out=None, residual=None, out_hl=None, relu_bits=None):
    """relu?(F * scale + shift + residual); out_hl: a second output in the hl format (cv_sp_affine_hl_f32)"""
    L = _lib.lib()
    dev = F.device
    if out is None:
        out = torch.empty_like(F)
    if out_hl is not None:
        with _on(dev):
            _lib.check(L.cv_sp_affine_hl_f32(_ptr(F), F.shape[0], F.shape[1], F.stride(0), _ptr(scale), _ptr(shift),
                                             _ptr(residual), residual.stride(0) if residual is not None else 0,
                                             1 if relu else 0, _ptr(out), out.stride(0), _ptr(out_hl), out_hl.stride(0),
                                             _ptr(relu_bits), range_flag(dev).data_ptr(), _stream(dev)), "cv_sp_affine_hl_f32")
        return out
    with _on(dev):
        _lib.check(L.cv_sp_affine_f32(_ptr(F), F.shape[0], F.shape[1], F.stride(0), _ptr(scale), _ptr(shift),
                                      _ptr(residual), residual.stride(0) if residual is not None else 0,
                                      1 if relu else 0, _ptr(out), out.stride(0), _stream(dev)),
                   "cv_sp_affine_f32")
    return out


class _BNTrainFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over feature rows on the HIP kernels (statistics, apply, backward), with the
    residual add and ReLU that follow it in BasicBlock folded into the same passes."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, residual, relu, want_hl=False):
        L = _lib.lib()
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        stats = torch.empty((4, c), dtype=torch.float32, device=dev)          # mean, var, scale, shift
        ws = torch.empty(int(L.cv_sp_bn_workspace_bytes(c)), dtype=torch.uint8, device=dev)
        with _on(dev):
            _lib.check(L.cv_sp_bn_stats_f32(_ptr(x), n, c, x.stride(0), _ptr(gamma), _ptr(beta), float(eps),
                                            float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(stats[0]),
                                            _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), _ptr(ws), ws.numel(),
                                            _stream(dev)), "cv_sp_bn_stats_f32")
        if residual is not None and residual.stride(1) != 1:
            residual = residual.contiguous()
        # want_hl: the output once more as fp16 pairs for the convolution that reads it (TRAIN_FWD_HL; c % 32 == 0)
        y_hl = torch.empty_like(x) if want_hl else None
        # ... and where its ReLU is open as one bit per element: what the backward reads instead of y (a 32nd of the bytes)
        bits = torch.empty((n, c // 32), dtype=torch.int32, device=dev) if (want_hl and relu) else None
        y = affine_forward(x, stats[2], stats[3], relu, residual=residual, out_hl=y_hl, relu_bits=bits)
        ctx.save_for_backward(x, gamma, stats, bits if bits is not None else (y if relu else None))
        ctx.bits = bits is not None
        ctx.eps = float(eps)
        ctx.has_res = residual is not None
        ctx.two = want_hl
        ctx.set_materialize_grads(False)         # (no zero tensor for the twin's "gradient": it was a fill of N x C per layer)
        if want_hl:
            _train_state.used_pairs = True
            ctx.mark_non_differentiable(y_hl)
            return y, y_hl
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        L = _lib.lib()
        x, gamma, stats, y = ctx.saved_tensors
        dy = dy.contiguous()
        n, c = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        dg = torch.empty((2, c), dtype=torch.float32, device=dev)
        dres = None
        if ctx.has_res and ctx.needs_input_grad[7]:
            dres = torch.empty_like(x) if y is not None else dy
        ws = torch.empty(int(L.cv_sp_bn_workspace_bytes(c)), dtype=torch.uint8, device=dev)
        bits = y if ctx.bits else None
        y_rows = None if ctx.bits else y
        slots = _bwd_ctx["slots"]
        slot, usable, dx_hl = None, False, None
        if slots is not None and c % 32 == 0 and x.stride(0) % 32 == 0 and x.stride(0) == dx.stride(0):
            slot, usable = slots.take(gamma.data_ptr())
            if slot is not None:
                # dx leaves with its hl twin for the input gradient of the convolution that produced x (usable from the layer's
                # second step on: the factor comes from the maximum the step before left in the slot)
                dx_hl = torch.empty_like(dx)
        with _on(dev):
            if dx_hl is not None or bits is not None:
                _lib.check(L.cv_sp_bn_backward_hl_f32(_ptr(x), _ptr(dy), _ptr(y_rows), n, c, x.stride(0), _ptr(stats[0]),
                                                      _ptr(stats[1]), ctx.eps, _ptr(gamma), _ptr(dg[0]), _ptr(dg[1]),
                                                      _ptr(dx), _ptr(dres) if y is not None else None, _ptr(ws), ws.numel(),
                                                      _ptr(dx_hl), _ptr(slot) if dx_hl is not None else None,
                                                      range_flag(dev).data_ptr(), _ptr(bits), _stream(dev)),
                           "cv_sp_bn_backward_hl_f32")
            else:
                _lib.check(L.cv_sp_bn_backward_f32(_ptr(x), _ptr(dy), _ptr(y), n, c, x.stride(0), _ptr(stats[0]),
                                                   _ptr(stats[1]), ctx.eps, _ptr(gamma), _ptr(dg[0]), _ptr(dg[1]),
                                                   _ptr(dx), _ptr(dres) if y is not None else None, _ptr(ws), ws.numel(),
                                                   _stream(dev)), "cv_sp_bn_backward_f32")
        if dx_hl is not None and usable:
            # (the entry holds dx itself: while it is in the table the allocator cannot hand dx's address to another gradient
            # tensor, so a twin whose consumer is not a _ConvFn - it then stays until the step's table is dropped - can never be
            # taken for a later tensor at the same address: ADVICE r5)
            _bwd_ctx["twins"][dx.data_ptr()] = (dx_hl, slot[8192:8193].view(torch.float32), dx.shape, dx)
        return dx, dg[0], dg[1], None, None, None, None, dres, None, None


# `num_batches_tracked += 1` of every BatchNorm of a training forward (62 one-element launches per MinkUNet34C step) as ONE
# multi-tensor add: a network forward opens the batch (batched_counter_updates), the modules append their counters
_nbt_pending = threading.local()


class batched_counter_updates:
    def __enter__(self):
        self.outer = getattr(_nbt_pending, "tensors", None)
        if self.outer is None:
            _nbt_pending.tensors = []
        return self

    def __exit__(self, *exc):
        if self.outer is None:
            pend, _nbt_pending.tensors = _nbt_pending.tensors, None
            if pend:
                with torch.no_grad():
                    torch._foreach_add_(pend, 1)
        return False


class MinkowskiBatchNorm(nn.Module):
    """``nn.BatchNorm1d`` over the feature rows; sub-module name ``bn`` (utils/resnet.py:115-116)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return self.forward_fused(x)

    def forward_fused(self, x, residual=None, relu=False):
        """relu?(bn(x) + residual) in one pass (BasicBlock's tail, resnet_block.py forward)."""
        bn = self.bn
        if self.training and bn.affine and bn.track_running_stats and bn.momentum is not None and x.F.shape[0] > 1:
            pend = getattr(_nbt_pending, "tensors", None)
            if pend is not None:
                pend.append(bn.num_batches_tracked)         # one multi-tensor add at the end of the network forward
            else:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
            if relu and train_forward_hl() and x.F.shape[1] % 32 == 0:
                # every convolution input of the network is the output of such a pass (or a concat of two): it leaves with
                # its hl twin
                y, y_hl = _BNTrainFn.apply(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                           residual, True, True)
                return x._like(y, F_hl=y_hl)
            return x._like(_BNTrainFn.apply(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                            bn.momentum, bn.eps, residual, bool(relu)))
        if self.training or torch.is_grad_enabled() and (x.F.requires_grad or residual is not None and
                                                         residual.requires_grad):
            y = self.bn(x.F)                      # unusual configurations / eval with autograd: torch's kernels
            if residual is not None:
                y = y + residual
            return x._like(torch.relu(y) if relu else y)
        scale, shift = bn_affine(self.bn)
        return x._like(affine_forward(x.F.contiguous(), scale, shift, relu, residual=residual))


class _SyncBNFn(torch.autograd.Function):
    """Training-mode BatchNorm over the rows of ALL ranks (scene-parallel DDP gives every rank its own scenes; the
    reference's statistics are over the batch of 3 scans on one GPU, config/config.yaml:15, train_joint.py:244-251).
    One all-reduce of [2C + 1] sums (count, sum x, sum x^2) in the forward, one of [2C] (sum dy, sum dy * xhat) in the
    backward; everything else is row-local.  Plain torch ops on purpose: the same code runs over gloo on CPU (the
    2-rank test) and over RCCL on the GPUs, and it is an option next to the per-GPU HIP BatchNorm, not the default."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, group):
        import torch.distributed as dist
        n, c = x.shape
        xd = x.double()
        pack = torch.cat([xd.new_full((1,), float(n)), xd.sum(0), (xd * xd).sum(0)])
        dist.all_reduce(pack, group=group)
        total = pack[0]
        mean = pack[1:1 + c] / total
        var = (pack[1 + c:] / total - mean * mean).clamp_(min=0.0)           # biased (normalisation) variance
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(momentum * mean.to(running_mean.dtype))
            unbiased = var * (total / (total - 1).clamp(min=1.0))
            running_var.mul_(1 - momentum).add_(momentum * unbiased.to(running_var.dtype))
        invstd = torch.rsqrt(var + eps)
        xhat = ((xd - mean) * invstd).to(x.dtype)
        ctx.save_for_backward(xhat, gamma, invstd.to(x.dtype))
        ctx.total = total
        ctx.group = group
        return xhat * gamma + beta

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        xhat, gamma, invstd = ctx.saved_tensors
        c = dy.shape[1]
        dyd = dy.double()
        sums = torch.cat([dyd.sum(0), (dyd * xhat.double()).sum(0)])
        dgamma, dbeta = sums[c:].to(dy.dtype), sums[:c].to(dy.dtype)          # local: DDP averages parameter grads
        dist.all_reduce(sums, group=ctx.group)
        m_dy = (sums[:c] / ctx.total).to(dy.dtype)
        m_dyx = (sums[c:] / ctx.total).to(dy.dtype)
        dx = (dy - m_dy - xhat * m_dyx) * (gamma * invstd)
        return dx, dgamma, dbeta, None, None, None, None, None


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """MinkowskiBatchNorm whose training-mode statistics span all ranks (ME.MinkowskiSyncBatchNorm); same ``bn``
    sub-module, same state-dict keys.  Eval mode and single-process runs are the base class."""
    process_group = None

    def forward_fused(self, x, residual=None, relu=False):
        import torch.distributed as dist
        bn = self.bn
        if not (self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1):
            return super().forward_fused(x, residual, relu)
        with torch.no_grad():
            bn.num_batches_tracked += 1
        y = _SyncBNFn.apply(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                            self.process_group)
        if residual is not None:
            y = y + residual
        return x._like(torch.relu(y) if relu else y)


def convert_sync_batchnorm(module, process_group=None):
    """every MinkowskiBatchNorm of ``module`` becomes a MinkowskiSyncBatchNorm in place (parameters, buffers and
    state-dict names untouched) - the counterpart of ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm"""
    for m in module.modules():
        if type(m) is MinkowskiBatchNorm:
            m.__class__ = MinkowskiSyncBatchNorm
            m.process_group = process_group
    return module


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        if x.F.requires_grad:
            return x._like(torch.relu(x.F))
        F = x.F.contiguous()
        return x._like(affine_forward(F, None, None, True, out=F if self.inplace else None))


def cat(*tensors):
    """Channel concat of tensors on the same coordinate map (utils/minkunet.py:153)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    t0 = tensors[0]
    for t in tensors[1:]:
        assert t.coordinate_manager is t0.coordinate_manager and t.tensor_stride == t0.tensor_stride
    hl = None
    if all(getattr(t, "F_hl", None) is not None and t.F.shape[1] % 32 == 0 for t in tensors):
        hl = torch.cat([t.F_hl for t in tensors], dim=1)       # 32-channel chunks keep their bytes: the concat of hl rows is hl
    return t0._like(torch.cat([t.F for t in tensors], dim=1), F_hl=hl)


from . import modules  # noqa: E402,F401
