"""The driver's eval launch line on an RCCL process group: `python -m torch.distributed.run
--nproc-per-node 1 ... bench.py --gpus 1 --steps 20 --warmup 5` with CV_DIST_FORCE=1 creates a one-rank `nccl` (= RCCL on
ROCm) group, so the default (eval) mode runs what an 8-GPU launch runs per rank: cvd.init("nccl"), the barriers around the
timed region, the max-reduction of the wall time.  One JSON line comes back with librccl mapped, and its parity object -
the HIP path of scene 0 against the CPU oracle - equals the plain single-process run's."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-reps", "1", "--train-steps", "0", "--min-warm-seconds", "0.3",
        "--measure-traffic", "0"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_eval_bench_on_a_one_rank_rccl_group_equals_the_plain_run(cuda, built_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = _one_line(subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *ARGS], cwd=ROOT, env=env,
                                     capture_output=True, text=True, timeout=1200))
    assert plain["collective"]["backend"] is None
    env["CV_DIST_FORCE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), *ARGS]
    rccl = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200))
    assert rccl["collective"] == {"backend": "nccl", "world_size": 1, "librccl_mapped": True,
                                  "librccl": rccl["collective"]["librccl"]} and rccl["collective"]["librccl"]
    assert rccl["n_gpus"] == 1 and rccl["steps"] == 20 and rccl["scaling"] == "weak"
    assert rccl["parity"] == plain["parity"]                 # the same scene through the same kernels: the same numbers
    for k in ("grid_shape_exact", "v_in_exact", "touched_cells_exact", "candidate_cells_exact", "box_count_exact",
              "classes_exact", "net_within_1e-4", "head_classes_exact"):
        assert rccl["parity"][k] is True, k
    assert rccl["value"] > 0 and plain["value"] > 0
