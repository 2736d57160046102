"""bench.py's launch path exactly as the driver starts it for N > 1 (python -m torch.distributed.run ... bench.py
--gpus N), without a GPU: argument parsing, env:// rendezvous on 127.0.0.1 (gloo stands in for RCCL), barriers, the
max-over-ranks reduction, rank 0 printing ONE JSON line - and a launch whose --gpus disagrees with the world size fails."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, gpus, extra=()):
    env = dict(os.environ, CV_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(gpus), "--steps", "5", "--warmup", "1", "--rendezvous-only", *extra]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


def test_two_rank_launch_rendezvous_and_single_json_line():
    r = _launch(2, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["rendezvous_only"]
    assert out["max_seconds"] >= 0.02                       # the max over ranks (rank 1 sleeps longer), not rank 0's time


def test_gpus_flag_must_equal_world_size():
    r = _launch(2, 3)
    assert r.returncode != 0
    assert "--gpus 3 but WORLD_SIZE is 2" in (r.stderr + r.stdout)


def test_single_process_default_is_one_gpu():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rendezvous-only"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rendezvous-only", "--gpus", "2"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0


def test_eight_rank_launch_and_scene_thread_cap():
    """the driver's 8-GPU line (one rank per GPU): rendezvous of eight ranks, one JSON line, and the scene threads of a
    rank capped by the cores the node has per rank (8 ranks x 4 threads on an 8-core box would be 32 runnable threads)"""
    r = _launch(8, 8, extra=("--streams", "4"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rendezvous_only"]
    cores = len(os.sched_getaffinity(0))
    assert out["scene_threads_per_rank"] == max(1, min(4, max(2, cores // 8)))
    assert out["max_seconds"] >= 0.08                       # rank 7 sleeps 80 ms: the max over ranks
