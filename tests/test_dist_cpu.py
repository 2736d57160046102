"""N>1 path of the scene-parallel benchmark on CPU: two gloo ranks (scenes shard
one per rank, no data-path collective; only the timing barrier and max/sum reductions)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world_size),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    from canonicalvoting_amd import dist as cvd
    from canonicalvoting_amd.synth import make_scene
    ws, r = cvd.init("gloo")
    assert (ws, r) == (world_size, rank)
    seeds = cvd.scene_seeds(rank, 2)
    scenes = [make_scene(s, n_points=500, res=0.06, room=(1.5, 0.9, 1.5), n_boxes=2, margin=0.5, box_scale=0.4)
              for s in seeds]
    cvd.barrier()
    my_time = 1.0 + rank                                  # pretend rank 1 is the slow one
    tmax = cvd.reduce_scalar(my_time, "max")
    total_pts = cvd.reduce_scalar(sum(len(s.coords) for s in scenes), "sum")
    cvd.barrier()
    q.put((rank, seeds, tmax, total_pts, cvd.throughput(2, ws, tmax), int(scenes[0].coords.sum())))
    cvd.finalize()


def test_two_rank_scene_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, t0, n0, v0, c0), (r1, s1, t1, n1, v1, c1) = out
    assert set(s0).isdisjoint(s1) and s0 == [0, 1] and s1 == [1000, 1001]
    assert t0 == t1 == 2.0                                # max over ranks
    assert n0 == n1 == 4 * 500                            # sum over ranks
    assert v0 == v1 == 2 * 2 / 2.0                        # whole-job scenes / max time
    assert c0 != c1                                       # different scenes on different ranks


def test_single_process_is_identity():
    from canonicalvoting_amd import dist as cvd
    os.environ.pop("WORLD_SIZE", None)
    assert cvd.reduce_scalar(3.5, "max") == 3.5 and cvd.throughput(10, 1, 2.0) == 5.0
    assert cvd.scene_seeds(3, 2) == [3000, 3001]


class _TinyNet(torch.nn.Module):
    """Linear -> BatchNorm over rows -> ReLU -> Linear: enough to push gradients through the statistics"""

    def __init__(self, sync):
        super().__init__()
        from canonicalvoting_amd import me as ME
        self.a = torch.nn.Linear(5, 16)
        self.norm = (ME.MinkowskiSyncBatchNorm if sync else ME.MinkowskiBatchNorm)(16)
        self.b = torch.nn.Linear(16, 3)
        self.sync = sync

    def forward(self, x):
        from canonicalvoting_amd import me as ME
        h = self.a(x)
        if self.sync:
            bn = self.norm.bn
            h = ME._SyncBNFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, None)
        else:
            h = self.norm.bn(h)
        return self.b(torch.relu(h))


def _syncbn_batch():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, 5, generator=g) * torch.tensor([1.0, 3.0, 0.2, 1.0, 5.0]) + torch.tensor([0.0, 2.0, -1.0, 0.5, 0.0])
    t = torch.randn(1000, 3, generator=g)
    return x, t


def _syncbn_worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world_size),
                      RANK=str(rank), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from canonicalvoting_amd import dist as cvd
    cvd.init("gloo")
    torch.manual_seed(0)
    net = _TinyNet(sync=True).train()
    ddp = DDP(net)
    x, t = _syncbn_batch()
    rows = slice(0, 700) if rank == 0 else slice(700, 1000)        # two scans on rank 0, one on rank 1
    xr = x[rows].clone().requires_grad_(True)
    # DDP averages parameter gradients over ranks: world * (sum over my rows) / N_total sums to the batch mean
    loss = world_size * ((ddp(xr) - t[rows]) ** 2).sum() / 1000.0
    loss.backward()
    # numpy, not tensors: a tensor in a queue is a shared-memory handle that dies with this process
    q.put((rank, {k: v.grad.numpy().copy() for k, v in net.named_parameters()}, xr.grad.numpy().copy(),
           net.norm.bn.running_mean.numpy().copy(), net.norm.bn.running_var.numpy().copy(), float(loss.detach())))
    cvd.finalize()


def test_sync_batchnorm_two_ranks_reproduce_the_single_process_batch_of_three():
    """train_joint.py:244-251 computes BatchNorm statistics over the 3 scans of a batch on one GPU; under
    scene-parallel DDP with --sync-bn two ranks holding 2 + 1 scans must give the same statistics, the same running
    buffers and - after DDP's gradient averaging - the same parameter and input gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=180) for _ in procs), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = _TinyNet(sync=False).train()
    x, t = _syncbn_batch()
    x = x.clone().requires_grad_(True)
    loss = ((ref(x) - t) ** 2).sum() / 1000.0
    loss.backward()
    T = torch.from_numpy
    (_, g0, dx0, rm0, rv0, l0), (_, g1, dx1, rm1, rv1, l1) = [
        (o[0], {k: T(v) for k, v in o[1].items()}, T(o[2]), T(o[3]), T(o[4]), o[5]) for o in out]
    assert abs((l0 + l1) / 2 - float(loss)) < 1e-5 * abs(float(loss))
    for k, p in ref.named_parameters():
        torch.testing.assert_close(g0[k], p.grad, rtol=1e-4, atol=1e-6)      # DDP left the same averaged gradient
        torch.testing.assert_close(g1[k], g0[k], rtol=0, atol=0)              # on both ranks
    torch.testing.assert_close(torch.cat([dx0, dx1]) / 2, x.grad, rtol=1e-4, atol=1e-7)
    for rm, rv in ((rm0, rv0), (rm1, rv1)):
        torch.testing.assert_close(rm, ref.norm.bn.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv, ref.norm.bn.running_var, rtol=1e-5, atol=1e-6)


def test_convert_sync_batchnorm_keeps_the_state_dict():
    from canonicalvoting_amd import me as ME
    from canonicalvoting_amd.minkunet import MinkUNet34C
    m = MinkUNet34C(3, 8)
    keys = list(m.state_dict())
    ME.convert_sync_batchnorm(m)
    assert list(m.state_dict()) == keys
    bns = [x for x in m.modules() if isinstance(x, ME.MinkowskiBatchNorm)]
    assert len(bns) == 62 and all(type(x) is ME.MinkowskiSyncBatchNorm for x in bns)


def _late_join_worker(rank, world_size, port, q):
    import torch.distributed as dist
    from canonicalvoting_amd import me as ME
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world_size)
    w = torch.nn.Parameter(torch.zeros(3, 4))
    with torch.no_grad():
        q.put((rank, ME._gradient_untouched_until_end(w)))
    dist.destroy_process_group()


def test_late_wgrad_join_only_when_nothing_can_touch_the_gradient():
    """me._ConvFn.backward may leave a layer's weight gradient running on its side stream until the end of the backward
    pass (CV_BACKWARD_OVERLAP=2) only if nothing reads or writes that gradient earlier: not with a gradient to
    accumulate into, a tensor hook, graph-building mode, a non-leaf weight - and never with more than one rank (DDP's
    reducer copies gradients into its buckets as they arrive)."""
    from canonicalvoting_amd import me as ME
    w = torch.nn.Parameter(torch.zeros(3, 4))
    derived = w * 1.0
    with torch.no_grad():                                   # the engine runs backward nodes with grad mode off
        assert ME._gradient_untouched_until_end(w)
        w.grad = torch.zeros_like(w)
        assert not ME._gradient_untouched_until_end(w)      # accumulation launches an add on the layer's stream
        w.grad = None
        h = w.register_hook(lambda g: g)
        assert not ME._gradient_untouched_until_end(w)
        h.remove()
        assert ME._gradient_untouched_until_end(w)
        assert not ME._gradient_untouched_until_end(derived)            # not a leaf: its gradient flows on
    assert not ME._gradient_untouched_until_end(w)          # create_graph / double backward
    port = _free_port()
    q = mp.get_context("spawn").SimpleQueue()
    mp.spawn(_late_join_worker, args=(2, port, q), nprocs=2, join=True)
    got = dict(q.get() for _ in range(2))
    assert got == {0: False, 1: False}


def _flag_worker(rank, world_size, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world_size), RANK=str(rank), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from canonicalvoting_amd import train
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    ring = train._FlagRing(torch.device("cpu"))
    log = []
    for step in range(8):
        if ring.noticed():
            log.append(("noticed", step))
            break
        # rank 1's forward leaves the fp16 range at step 3 (and its flag is sticky); rank 0's never does
        local = 1.0 if (rank == 1 and step >= 3) else 0.0
        found = ring.push(torch.tensor(local), dist.group.WORLD)
        log.append(("skip" if float(found) else "step", step))
    q.put((rank, log))
    dist.destroy_process_group()


def test_two_ranks_act_on_the_same_range_flag_at_the_same_step():
    """train.train_step's range flag under DDP.  One rank's forward overflows at step 3: after the
    all-reduce (MAX) BOTH ranks skip the update of steps 3 and 4 (found_inf), and both notice at step 3 + FLAG_LAG - the
    same call - where each restores, redoes and switches to the triples; nobody is left in a collective alone."""
    from canonicalvoting_amd import train
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [("step", 0), ("step", 1), ("step", 2)] + [("skip", 3 + i) for i in range(train.FLAG_LAG)] + [("noticed", 3 + train.FLAG_LAG)]
    assert out[0] == out[1] == want, out
    # one process, no group: the same lag on the local flag
    ring = train._FlagRing(torch.device("cpu"))
    seen = []
    for step in range(6):
        seen.append(ring.noticed())
        ring.push(torch.tensor(1.0 if step >= 1 else 0.0))
    assert seen == [False, False, False, True, True, True]
    ring = train._FlagRing(torch.device("cpu"))
    assert ring.push(None) is None and not ring.noticed() and not ring.noticed()       # a step without fp16 pairs stores nothing
