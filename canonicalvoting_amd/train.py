"""Training step of the joint model as functions (train_joint.py:244-288):

    joint_loss          head gather by ground-truth class + masked MSE(xyz) + MSE(log scale) + CE(class)
                        (train_joint.py:253-283; defaults from config/config.yaml)
    separate_loss       the per-category model's loss with the minimum over symmetry-equivalent poses
                        (train_separate.py:247-287)
    train_step          zero_grad -> forward -> loss -> backward -> optimizer.step (:246-288)
    adjust_learning_rate step LR decay (:128-138, config.yaml:32-36)
    make_ddp            scene-parallel data parallelism: torch DDP (bucketed gradient all-reduce over
                        RCCL, overlapped with backward) around the model (the reference
                        itself is single-GPU)

The sparse convolutions run forward / input-gradient / weight-gradient on the HIP kernels
(canonicalvoting_amd.me._ConvFn), batch-statistics BatchNorm with the residual add and ReLU folded in on
bn_col_reduce4 / bn_backward_apply4 (me._BNTrainFn); the losses are torch ops.
The reference's stale-loss bug (`losses` persists across iterations) is not
reproduced: a batch without object points contributes only the classification loss.
"""
import torch
import torch.nn.functional as F

from . import me as ME


def joint_loss(out_feats, xyz_labels, scale_labels, class_labels, nclasses=9, log_scale=True,
               xyz_factor=1.0, scale_factor=1.0, xyz_component_weights=(1.0, 1.0, 1.0)):
    """out_feats [N, 7*nclasses+1]; labels as utils/dataloader.py:170-188 (class 9 / -1 = background)."""
    labels = class_labels.long()
    idx = labels.clamp(min=0)
    idx = torch.where(idx == nclasses, torch.zeros_like(idx), idx)             # :254-256
    g = idx[:, None, None].expand(-1, 1, 3)
    out_xyz = torch.gather(out_feats[:, :3 * nclasses].reshape(-1, nclasses, 3), 1, g)[:, 0]       # :257
    out_scale = torch.gather(out_feats[:, 3 * nclasses:6 * nclasses].reshape(-1, nclasses, 3), 1, g)[:, 0]
    out_class = out_feats[:, 6 * nclasses:]
    mask = (labels < nclasses) & (labels >= 0)                                  # :261
    w = torch.as_tensor(xyz_component_weights, dtype=out_feats.dtype, device=out_feats.device)
    # :262-272 without a host wait.  The reference indexes with the boolean mask (`if torch.any(mask)`, `x[mask]`): every such
    # line makes the host wait for the forward, and the backward is only queued once the chip has run dry (3-4 ms of a 27 ms
    # step, profiles/r5/train_gaps.txt).  The same means as sums over all rows times the mask, divided by the object rows on
    # the device: equal up to summation order; no object row: both terms are zero, as the reference's `if` leaves them.
    # Background rows are SELECTED away in front of the arithmetic (prediction and target both become 0 there), not multiplied by
    # zero behind it: a non-finite prediction in a row the reference never touches (:262-272 index with the mask) reaches
    # neither the loss nor - through 0 * inf in a backward formula - a gradient.
    m = mask[:, None]
    cnt = mask.sum()
    denom = (cnt * 3).clamp(min=1).to(out_feats.dtype)
    zero = torch.zeros((), dtype=out_feats.dtype, device=out_feats.device)
    safe_scale = torch.where(m, scale_labels, torch.ones_like(scale_labels))
    tgt_scale = torch.where(m, torch.log(safe_scale) if log_scale else safe_scale, zero)       # :266-269
    tgt_xyz = torch.where(m, xyz_labels, zero)
    out_scale = torch.where(m, out_scale, zero)
    out_xyz = torch.where(m, out_xyz, zero)
    losses = {"loss_scale": torch.sum((out_scale - tgt_scale) ** 2 * w) / denom * scale_factor,
              "loss_xyz": torch.sum((out_xyz - tgt_xyz) ** 2 * w) / denom * xyz_factor}
    # :273; the reference's labels are 0..9 (9 = background, utils/dataloader.py:172) - a negative label would make its
    # CrossEntropyLoss raise, here it counts as background
    losses["loss_class"] = F.cross_entropy(out_class, torch.where(labels < 0, torch.full_like(labels, nclasses), labels))
    return sum(losses.values()), losses


def separate_loss(out_feats, xyz_labels, scale_labels, obj_labels, coords4=None, log_scale=True, xyz_factor=1.0,
                  scale_factor=1.0, xyz_component_weights=(1.0, 1.0, 1.0), reference_indexing=True):
    """Loss of the per-category model (train_separate.py:247-287): out_feats [N, 8] = xyz, (log) scale, 2 objectness
    logits; ``xyz_labels`` per scan a list of [rows, [xyz under each symmetry-equivalent pose]]
    (data.collate_fn_separate); the coordinate loss of a model is the MINIMUM over its poses (:271-275), averaged
    over models.  ``reference_indexing``: the reference indexes the BATCH output with each scan's own row numbers
    (:270, `batch_idx` is computed and never used), which is only right for the first scan of a batch; True keeps
    that, False offsets the rows of scan j by the rows of scans < j (needs ``coords4``)."""
    obj = obj_labels.long()
    mask = obj == 1
    w = torch.as_tensor(xyz_component_weights, dtype=out_feats.dtype, device=out_feats.device)
    losses = {"loss_obj": F.cross_entropy(out_feats[:, 6:8], obj)}
    tgt = torch.log(scale_labels[mask]) if log_scale else scale_labels[mask]
    losses["loss_scale"] = torch.mean((out_feats[:, 3:6][mask] - tgt) ** 2 * w) * scale_factor
    out_xyz = out_feats[:, :3]
    first_row = None
    if not reference_indexing:
        counts = torch.bincount(coords4[:, 0].long(), minlength=len(xyz_labels))
        first_row = torch.cumsum(counts, 0) - counts
    per_model = []
    for j, per_scan in enumerate(xyz_labels):
        for rows, xyzs in per_scan:
            rows = rows.to(out_feats.device)
            if first_row is not None:
                rows = rows + first_row[j]
            pred = out_xyz[rows]
            per_model.append(torch.stack([torch.mean((pred - x.to(out_feats.device)) ** 2 * w) for x in xyzs]).min())
    losses["loss_xyz"] = torch.mean(torch.stack(per_model)) * xyz_factor
    return sum(losses.values()), losses


def _skips_on_device(optimizer):
    """a fused torch Adam / AdamW takes `found_inf` (what torch.cuda.amp.GradScaler hands it): the update is skipped on the device"""
    return isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)) and all(g.get("fused") for g in optimizer.param_groups)


FLAG_LAG = 2        # train_step reads the range flag of the step this many calls ago (all ranks the same one)


def _step_group(model):
    """the process group whose ranks step together with this model (DDP), or None"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = getattr(model, "process_group", None)
    if group is None and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
        return None
    return group if dist.get_world_size(group) > 1 else None


class _FlagRing:
    """The range flag of the step FLAG_LAG calls ago - the SAME step on every rank, after an all-reduce (MAX) over the group:
    ranks that step together act on one value at one step (skip / restore / redo / switch to the triples together; a rank
    acting on its own flag, or on a live peek that lands at another step, leaves the others with diverged parameters or in a
    collective nobody joins: ADVICE r5).  push() stores this step's flag behind the step's work (pinned memory + event, no
    host wait); noticed() reads the oldest stored one, whose event passed long ago."""

    def __init__(self, dev, lag=FLAG_LAG):
        self.dev, self.lag, self.k, self.slots = dev, lag, 0, [None] * lag

    def noticed(self):
        slot = self.slots[self.k % self.lag]
        if slot is None or not slot["valid"]:
            return False
        if slot["event"] is not None:
            slot["event"].synchronize()
        return float(slot["host"][0]) != 0.0

    def push(self, found, group=None):
        """found: float scalar tensor on the step's device (None: the step used no fp16 pairs); returns the group's value"""
        i = self.k % self.lag
        self.k += 1
        slot = self.slots[i]
        if slot is None:
            cuda = self.dev.type == "cuda"
            host = torch.zeros(1, dtype=torch.float32)
            slot = self.slots[i] = {"host": host.pin_memory() if cuda else host, "event": torch.cuda.Event() if cuda else None,
                                    "valid": False}
        slot["valid"] = found is not None
        if found is None:
            return None
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(found, op=dist.ReduceOp.MAX, group=group)
        slot["host"].copy_(found.reshape(1), non_blocking=True)
        if slot["event"] is not None:
            slot["event"].record(torch.cuda.current_stream(self.dev))
        return found


def _bn_state(model):
    """(the BatchNorm running statistics and counters of `model`, a snapshot buffer for each that lives with the model)"""
    bufs = [b for m in model.modules() for b in (getattr(m, "running_mean", None), getattr(m, "running_var", None),
                                                 getattr(m, "num_batches_tracked", None)) if b is not None]
    saved = model.__dict__.get("_bn_snapshot")
    if saved is None or len(saved) != len(bufs) or any(a.shape != b.shape or a.device != b.device or a.dtype != b.dtype
                                                       for a, b in zip(saved, bufs)):
        saved = model.__dict__["_bn_snapshot"] = [torch.empty_like(b) for b in bufs]
    return bufs, saved


def train_step(model, optimizer, coords4, feats, xyz_labels, scale_labels, class_labels, **loss_kw):
    """one iteration of train_joint.py:246-288; feats already recentred (:248-249).

    The forward multiplies fp16 pairs on the eval path's kernels (ME.TRAIN_FWD_HL); an activation beyond the fp16 range - never
    behind a healthy BatchNorm - raises the range flag.  With a fused Adam the flag rides to the optimizer as `found_inf`: the
    update of such a step is skipped ON THE DEVICE (no host wait in the step).  The host reads the flag of the step FLAG_LAG
    calls ago (pinned memory + an event that has long passed): it then restores the BatchNorm running statistics and counters
    from a snapshot that a device-side guard has kept at the state before the FIRST flagged step (cv_sp_copy_unless_flag: the
    flagged batches are dropped as a whole - no update, no statistics, exactly as if they had not been drawn), counts the
    fallback (model.train_range_fallbacks) and runs this and the following steps on the bf16 triples.  Other optimizers: the host
    waits for the flag and redoes the step on the triples before the optimizer sees a gradient.
    Under DDP the flag is all-reduced (MAX) over the ranks of the model's process group before anybody acts on it, and the
    lagged read looks at the same step on every rank: all ranks skip, restore, redo and switch to the triples together."""
    dev = feats.device
    group = _step_group(model)

    def fwd_bwd():
        optimizer.zero_grad(set_to_none=True)
        x = ME.SparseTensor(feats, coords4, device=dev)
        out = model(x)
        loss, parts = joint_loss(out.F, xyz_labels, scale_labels, class_labels, **loss_kw)
        loss.backward()
        return loss, parts

    def on_triples(fn):
        prev, ME.TRAIN_FWD_PIECES = ME.TRAIN_FWD_PIECES, 3
        prev_hl, ME.TRAIN_FWD_HL = ME.TRAIN_FWD_HL, 0
        try:
            return fn()
        finally:
            ME.TRAIN_FWD_PIECES = prev
            ME.TRAIN_FWD_HL = prev_hl

    if not ME.train_uses_pairs() or model.__dict__.get("_pairs_off", False):
        loss, parts = on_triples(fwd_bwd) if ME.train_uses_pairs() else fwd_bwd()
        optimizer.step()
        return loss.detach(), {k: v.detach() for k, v in parts.items()}

    bufs, saved = _bn_state(model)
    if _skips_on_device(optimizer):
        ring = model.__dict__.get("_flag_ring")
        if ring is None or ring.dev != dev:
            ring = model.__dict__["_flag_ring"] = _FlagRing(dev)
        if ring.noticed():
            # the step FLAG_LAG calls ago left the fp16 range (on this rank or, under DDP, on any rank): its update and that
            # of every step queued since was skipped on the device, and the BatchNorm snapshot still holds the statistics and
            # counters of before it
            torch.cuda.current_stream(dev).synchronize()
            with torch.no_grad():
                torch._foreach_copy_(bufs, saved)
            ME.range_flag(dev).zero_()
            model.train_range_fallbacks = getattr(model, "train_range_fallbacks", 0) + 1
            model.__dict__["_pairs_off"] = True
            model.__dict__.pop("_flag_ring", None)
            return train_step(model, optimizer, coords4, feats, xyz_labels, scale_labels, class_labels, **loss_kw)
        # BatchNorm statistics of before this step, kept ON THE DEVICE only while no step has raised the flag
        ME.copy_unless_flag(bufs, saved, dev, ME.range_flag(dev))
        with ME.pair_scale_hints(model):
            loss, parts = fwd_bwd()
        found = ring.push(ME.training_range_flag_device(dev), group)
        if found is not None and group is not None:
            # the local flag follows the group's: every rank's later steps skip too, and every rank's snapshot guard closes
            ME.range_flag(dev).copy_(found.reshape(1).to(torch.int32), non_blocking=True)
        optimizer.found_inf = found
        try:
            optimizer.step()
        finally:
            optimizer.found_inf = None
        return loss.detach(), {k: v.detach() for k, v in parts.items()}

    # the step may have to be redone: keep the BatchNorm running statistics of before the step, so that the redo does not count
    # the batch twice (one momentum update and one num_batches_tracked increment per iteration, as train_joint.py:250-283 gives)
    # (one multi-tensor copy into buffers that live with the model: 186 one-element clones per step were 186 launches)
    with torch.no_grad():
        torch._foreach_copy_(saved, bufs)
    with ME.pair_scale_hints(model):
        loss, parts = fwd_bwd()
    if ME.training_forward_left_fp16_range(dev, group):
        with torch.no_grad():
            torch._foreach_copy_(bufs, saved)
        loss, parts = on_triples(fwd_bwd)
        model.train_range_fallbacks = getattr(model, "train_range_fallbacks", 0) + 1
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in parts.items()}


def train_step_separate(model, optimizer, coords4, feats, xyz_labels, scale_labels, obj_labels, **loss_kw):
    """one iteration of train_separate.py:236-292 (8-channel per-category model); feats already recentred (:241-242).
    Returns None for a batch without object points, which the reference skips (:238-240)."""
    if not bool((obj_labels == 1).any()):
        return None
    optimizer.zero_grad(set_to_none=True)
    out = model(ME.SparseTensor(feats, coords4, device=feats.device))
    dev = feats.device
    loss, parts = separate_loss(out.F, xyz_labels, scale_labels.to(dev), obj_labels.to(dev), coords4=coords4, **loss_kw)
    loss.backward()
    optimizer.step()
    return loss.detach(), {k: v.detach() for k, v in parts.items()}


def adjust_learning_rate(optimizer, epoch, base_lr=1e-3, decay_steps=(80, 120, 160), decay_rates=(0.1, 0.1, 0.1)):
    lr = base_lr
    for s, r in zip(decay_steps, decay_rates):
        if epoch >= s:
            lr *= r
    for g in optimizer.param_groups:
        g["lr"] = lr
    return lr


def bn_momentum(epoch, decay_step, decay_rate, init=0.5, floor=0.001):
    """train_joint.py:200-201,224: max(0.5 * rate ** (epoch // step), 0.001)"""
    return max(init * decay_rate ** int(epoch / decay_step), floor)


def set_bn_momentum(model, momentum, effective=False):
    """BNMomentumScheduler.step (train_joint.py:86-125).  The reference's setter assigns ``m.momentum`` on every
    ``ME.MinkowskiBatchNorm`` (:94-98) - the WRAPPER, whose forward runs its ``.bn`` (an nn.BatchNorm1d created with
    momentum 0.1): the running statistics of the reference therefore keep updating with 0.1 whatever the schedule
    says.  Default here = the same observable behaviour (the attribute is set on the wrapper, ``.bn`` is untouched);
    ``effective=True`` applies the schedule to ``.bn.momentum`` as its authors presumably intended.  Returns the
    number of modules touched."""
    k = 0
    for m in model.modules():
        if isinstance(m, ME.MinkowskiBatchNorm):
            m.momentum = momentum
            if effective:
                m.bn.momentum = momentum
            k += 1
    return k


def make_optimizer(model, lr=1e-3, weight_decay=0.0):
    """torch.optim.Adam as train_joint.py:219-223 (fused multi-tensor implementation on the GPU)."""
    params = [p for p in model.parameters() if p.requires_grad]
    fused = all(p.is_cuda for p in params)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, fused=fused)


def make_ddp(model, device, bucket_cap_mb=48):
    """One process per GPU; 37.9 M fp32 gradients = 151 MB per step in ~48 MB buckets so the
    all-reduce of the decoder's gradients overlaps the encoder's backward."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    return DDP(model, device_ids=[device.index], bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)
