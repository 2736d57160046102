"""Scene-parallel sharding helpers: scenes are independent units, rank r owns its own
scenes and there is NO data-path collective; torch.distributed (RCCL on GPUs, gloo in CPU tests) is
only used to align the timed region and to reduce the wall time / detection counts."""
import os

import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(
        os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1 (env:// rendezvous)."""
    ws, rank, _ = world()
    # CV_DIST_BACKEND=gloo: the launch path (env rendezvous, barriers, max / sum reductions) on a box whose ranks
    # share one GPU, where RCCL refuses to start (profiles/two_ranks_one_gpu.sh); never set in production
    backend = os.environ.get("CV_DIST_BACKEND", backend)
    # CV_DIST_FORCE=1: create the group for a single rank too (RCCL with one rank: `bench.py --mode train` under
    # `torch.distributed.run --nproc-per-node 1` then runs the DDP path a multi-GPU node runs)
    force = os.environ.get("CV_DIST_FORCE", "0") == "1" and "MASTER_ADDR" in os.environ
    if (ws > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return ws, rank


def scene_seeds(rank, scenes_per_rank):
    """Seeds of the scenes a rank owns: disjoint across ranks (scene i of rank r = r*1000 + i)."""
    assert scenes_per_rank <= 1000
    return [rank * 1000 + i for i in range(scenes_per_rank)]


def barrier(device=None):
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def reduce_scalar(value, op="max", device=None):
    """max / sum of a python float over ranks (identity for a single process)."""
    if not dist.is_initialized():
        return float(value)
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def throughput(steps_per_rank, world_size, max_seconds):
    """whole-job scenes/s: every rank processes steps_per_rank scenes in the max-over-ranks time"""
    return steps_per_rank * world_size / max_seconds


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
