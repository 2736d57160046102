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
