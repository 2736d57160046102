#!/usr/bin/env python
"""Headline benchmark: scenes/sec on 80k-point synthetic scans (BASELINE.json metric, config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one synthetic 80k-voxel scene through the eval_joint.py path (eval_joint.py:163-280) with its inputs
already resident in HBM: coordinate manager + kernel maps, sparse MinkUNet34C forward, head split, vote, greedy
decode with the back-projection check, per-class NMS.  Scenes are independent, so N ranks run N scenes per step with
no data-path collective ("scaling": "weak"); value = scenes all ranks processed / max-over-ranks time.

Predictions (--predictions): the network is random-init (no checkpoint offline), and a random-init network never
crosses thresh_high, which would leave decode / back-projection / NMS with nothing to do.  The default "teacher"
therefore runs the network and the head kernel of every step as eval_joint.py does (timed), and then feeds vote + decode
with predictions synthesised from the scene's labels (SURVEY 8d recipe: the peaked vote maps a trained network
produces, ~12 boxes per scene).  "network" feeds the network's own output (detections_per_scene 0).

Scenes in flight: --streams S (default 7) host threads, each with its own HIP stream, take the K steps from one
shared counter.  The threads are created, bound to their streams and parked on a barrier BEFORE the timed region
starts.  Per-scene work and results are unchanged (tests assert bit-identity with the one-at-a-time path).

The JSON line also carries
  roofline      the vote op (every launch of cv_hv_forward_f32) timed with HIP events on the stream it runs on,
                INSIDE the timed region, priced with the algorithmic bytes of SURVEY.md 8d:
                B_vote = 40 N + 192 V_in + 68 G.  With S > 1 the events include the stretch from co-running scenes;
                `isolated_*` give the same op from a one-scene-in-flight pass after the timed region.
  stage_ms      per-stage event times inside the timed region (and `stage_ms_isolated` from that second pass): DEVICE times -
                every boundary event sits behind the stage's last launch (the decode's in front of its host wait)
  parity        match flags of the SAME run against the CPU oracle on one scene, UNDER THE LAUNCH SIZING OF THE TIMED REGION
                (`parity.config`): grid shape, in-bounds vote count, candidate cells, box count, classes exact; network max
                abs error.  `parity_one_in_flight`: the same under the library's one-scene sizing, plus whether the two runs
                agree bit for bit on the grids and every integer output.
  launch sizing the three choices that depend on the scenes in flight (`config.conv_split_target`, `vote_part_records`,
                `masked_min_rows`) travel with every call (pipeline.ScenePolicy -> cv_scene_desc): nothing process-wide
  steady_state  side field (one rank, --steps below --steady-steps): the same scene threads, streams and launch policy over 160
                more steps right after the timed region - the driver's 20-step region spends ~9 % of its 37 ms filling and
                draining a seven-deep pipeline; never `value`
  cu_busy_in_flight, roofline_conv.mfma_busy_in_flight, in_flight_counters
                counters of the regime that is timed, collected offline (dispatch counters serialise the kernels) by
                profiles/in_flight_counters.sh and read from profiles/r6/in_flight_counters.json (labelled as such)
  cpu_baseline  the CPU oracle (oracle/, a port - the reference has no CPU path) on a bounded sample of the same
                scenes, rank 0 only: best of --cpu-reps after one warm-up on all host threads, plus one 1-thread run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from canonicalvoting_amd import _lib, decode, hv_cuda, pipeline  # noqa: E402
from canonicalvoting_amd import dist as cvd  # noqa: E402
from canonicalvoting_amd import me as ME  # noqa: E402
from canonicalvoting_amd.hough import HoughVoting  # noqa: E402
from canonicalvoting_amd.minkunet import MinkUNet34C  # noqa: E402
from canonicalvoting_amd.synth import make_scene, synth_predictions  # noqa: E402

N_POINTS = 80000
NUM_ROTS = 120
RES = 0.03
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="must equal WORLD_SIZE (one rank per GPU)")
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--scenes", type=int, default=4, help="distinct resident scenes per rank")
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--algo", type=int, default=0, help="vote algorithm: 0 auto, 1 direct, 2 tiles")
    ap.add_argument("--cpu-scenes", type=int, default=1, help="scenes timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-reps", type=int, default=None,
                    help="repetitions of the CPU oracle at its best torch thread count (after a warm-up and one run per "
                         "thread count; best is reported).  Default 3, or 1 with WORLD_SIZE > 1 - the other ranks wait "
                         "for rank 0 behind the final barrier")
    ap.add_argument("--stage", default="full", choices=["vote_decode", "full"],
                    help="full = MinkUNet34C forward + head + vote + decode + NMS (eval_joint.py path)")
    ap.add_argument("--min-warm-seconds", type=float, default=1.5,
                    help="the untimed warm-up lasts at least this long (sustained work before the clock starts); "
                         "--warmup is a minimum number of steps, not the whole warm-up")
    ap.add_argument("--streams", type=int, default=7,
                    help="scenes in flight per GPU: S host threads, each with its own HIP stream, take the steps "
                         "from one shared counter (scenes are independent; fills the launch tails and host syncs of one scene "
                         "with the kernels of another).  7 and 8 give the same rate over 240 steps (575-577 scenes/s); a "
                         "20-step region ends with fewer idle threads at 7 (rounds of 7 + 7 + 6 against 8 + 8 + 4: 528-532 "
                         "against 503-516 scenes/s, profiles/r4/streams_20steps.txt, hwq_20steps.txt)")
    ap.add_argument("--switch-interval", type=float, default=0.0005, help="sys.setswitchinterval for the scene threads")
    ap.add_argument("--event-every", type=int, default=1,
                    help="steps of the timed region that carry HIP events (stage boundaries + the vote kernel): every K-th "
                         "(1: all).  An event is a marker packet in the stream's hardware queue; the stage times and the "
                         "`roofline` of the line come from the steps that have them")
    ap.add_argument("--kernel-events", type=int, default=1,
                    help="0: no cv_hv_set_kernel_events pair around the vote kernel (`roofline` then prices the whole op)")
    ap.add_argument("--split-target", type=int, default=-1,
                    help="cv_sp_set_split_target for the timed region: workgroups a split convolution launch aims at. "
                         "-1 (default): 256 from four scenes in flight (the other scenes fill the chip, the partial "
                         "tiles only cost traffic), else the library's 512; the one-scene-in-flight side pass always "
                         "runs on the library default")
    ap.add_argument("--vote-part-records", type=int, default=-1,
                    help="cv_hv_set_part_records for the timed region: records of a plane's two y-bins one vote workgroup takes. "
                         "-1 (default): 12288 from four scenes in flight (fewer, longer workgroups and less merge traffic while "
                         "the other scenes fill the chip), else the library's 4096; the one-scene-in-flight side pass always "
                         "runs on the library default.  The grids are the same bits under every setting")
    ap.add_argument("--masked-min-rows", type=int, default=-1,
                    help="rows from which a level's 3x3x3 convolutions run mask-sorted (cv_scene_desc.masked_min_rows) in the timed "
                         "region.  -1 (default): 8192 from four scenes in flight (the ts4 level too: fewer live units per workgroup, "
                         "one more set of partial tiles - pays when other scenes fill the chip), else the library's 16384 (best for "
                         "one scene at a time); the one-scene-in-flight side pass always runs on the library default")
    ap.add_argument("--stagger-us", type=float, default=400.0,
                    help="scene thread i takes its first timed step i x this many microseconds after the clock started: scenes that "
                         "start together stay in the same stage (all in the convolutions, then all in the vote) and share the chip "
                         "worse than scenes a fraction of a scene apart")
    ap.add_argument("--measure-traffic", type=int, default=-1,
                    help="roofline.traffic from the PMC counters IN this run (rank 0, single process, default 80k workload): "
                         "two child runs of `bench.py --stage vote_decode` under rocprofv3 (--kernel-trace --pmc FETCH_SIZE, "
                         "then WRITE_SIZE: separate passes), after the timed region.  -1 = when rocprofv3 is on PATH; 0 = "
                         "never (the value of profiles/r*/vote_hbm_traffic.json is reported, labelled as read from a file)")
    ap.add_argument("--tail-priority", type=int, default=0,
                    help="the last N steps of the timed region run on high-priority HIP streams (one per scene thread, "
                         "hipStreamCreateWithPriority): the scenes that start last have the most work left when the ticket "
                         "counter runs out, so favouring them shortens the drain of a short region (longest remaining work "
                         "first); fill and drain stay inside the timed region, results are bit-identical.  0 = off")
    ap.add_argument("--steady-steps", type=int, default=160,
                    help="side field `steady_state`: when --steps is smaller than this, the scene threads run this many more steps "
                         "after the timed region (one rank only); 0 = off")
    ap.add_argument("--xcd-partition", type=int, default=0,
                    help="experiment: the scene streams are created with CU masks (hipExtStreamCreateWithCUMask) of this many XCDs "
                         "each (1, 2 or 4 of the 8; stream i takes group i mod (8 / N)): a scene's kernels stay on its XCDs' CUs and "
                         "L2s.  0 (default): every stream may use the whole chip")
    ap.add_argument("--mode", default="eval", choices=["eval", "train", "separate"],
                    help="eval (default, the BASELINE metric): eval_joint.py path.  train: train_joint.py step "
                         "(fwd + bwd + Adam, fp32) on --train-batch scenes per GPU-step, DDP gradient all-reduce over "
                         "RCCL when launched on several GPUs (BASELINE configs 3-4; a side measurement, not the "
                         "headline metric)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="product precision of the sparse convolutions.  f32 (default, the parity path): fp32-level "
                         "products on the 16-bit matrix cores (fp16 pairs / bf16 triples).  bf16: operands rounded to "
                         "bf16, one product, fp32 accumulation and storage (BASELINE configs 3-4 name bf16 for "
                         "training; outside the 1e-4 parity bar, reported as dtype bf16)")
    ap.add_argument("--train-batch", type=int, default=3, help="scenes per GPU-step in --mode train (config.yaml:15)")
    ap.add_argument("--train-steps", type=int, default=None,
                    help="eval mode, rank 0: train_joint.py steps timed AFTER the timed region for the `train_step_ms` side "
                         "field (0 = skip).  Default 5, or 0 with WORLD_SIZE > 1")
    ap.add_argument("--sync-bn", action="store_true", help="--mode train: BatchNorm statistics over all ranks' rows "
                                                           "(the reference's batch-of-3 semantics under scene-parallel DDP)")
    ap.add_argument("--large", action="store_true",
                    help="BASELINE config 5 shaped scenes: 9x3x9 m room, 40 boxes (use with --points 300000)")
    ap.add_argument("--predictions", default="teacher", choices=["teacher", "network"],
                    help="what vote + decode are fed with (see the module docstring); the network forward and the head "
                         "kernel run and are timed either way")
    ap.add_argument("--teacher-forced", action="store_true", help="same as --predictions teacher (kept for old scripts)")
    ap.add_argument("--scene-call", default="c", choices=["c", "py"],
                    help="c (default): one cv_detect_scene_f32 call per scene (plan -> network -> head -> vote -> decode -> NMS "
                         "inside the library, the GIL released for the whole scene); py: the same entry points issued one "
                         "by one from Python (the path of rounds 1-3; bit-identical results)")
    ap.add_argument("--adaptive-split", type=int, default=0,
                    help="--scene-call c without an explicit --split-target: every scene picks its split target by the scenes "
                         "inside cv_detect_scene_f32 when it starts (cv_scene_desc.adaptive_split); 0: the fixed setting")
    ap.add_argument("--ablate", default="", help="timing ablations, WRONG results, never for a reported number: comma list of "
                                                  "`finish` (cv_sp_set_ablation bit 0, switched on after the warm-up), `novote`, "
                                                  "`nodecode` (the stage is skipped in every step), `noplan` (the coordinate plan of a resident "
                                                  "scene is built once and reused)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch-path check without a GPU: parse, rendezvous (CV_DIST_BACKEND=gloo on CPU), barrier, "
                         "max-reduce, print the JSON skeleton")
    a = ap.parse_args()
    if a.teacher_forced:
        a.predictions = "teacher"
    return a


def scene_threads(requested):
    """scene threads (= scenes in flight) of this rank: what was asked for, but never more than the host cores the rank
    can run on (8 ranks x S threads share one node's cores; at least 2 so that host syncs of one scene overlap another)"""
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(max(1, requested), max(2, cores // max(1, local_world))))


def xcd_streams(dev, count, xcds_per_stream):
    """streams whose kernels run on a group of XCDs only: hipExtStreamCreateWithCUMask.  The CU mask of a multi-XCD device
    is interleaved - bit k belongs to XCD k mod 8 - so a group of XCDs is one byte pattern repeated over the 256 bits."""
    import ctypes
    assert xcds_per_stream in (1, 2, 4)
    hip = ctypes.CDLL("libamdhip64.so")
    groups = 8 // xcds_per_stream
    out = []
    for i in range(count):
        g = i % groups
        byte = 0
        for x in range(g * xcds_per_stream, (g + 1) * xcds_per_stream):
            byte |= 1 << x
        word = byte | (byte << 8) | (byte << 16) | (byte << 24)
        mask = (ctypes.c_uint32 * 8)(*([word] * 8))
        st = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, mask)
        assert rc == 0 and st.value, "hipExtStreamCreateWithCUMask failed (%d)" % rc
        out.append(torch.cuda.ExternalStream(st.value, device=dev))
    return out


def check_world(a):
    """--gpus is the contract's statement of the world size: a launch line whose --gpus and --nproc-per-node disagree
    would report a whole-job value for a job that was not run"""
    world, rank, local = cvd.world()
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with python -m torch.distributed.run "
                         "--nproc-per-node %d ... bench.py --gpus %d)" % (a.gpus, world, a.gpus, a.gpus))
    # rank 0's side legs run while the other ranks idle: a cheap tail on a multi-GPU launch unless the caller asks otherwise
    if a.cpu_reps is None:
        a.cpu_reps = 3 if world == 1 else 1
    if a.train_steps is None:
        a.train_steps = 5 if world == 1 else 0
    return world, rank, local


def measure_vote_traffic(a):
    """HBM bytes per launch of hv_fwd_tiles from the PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE in SEPARATE rocprofv3 passes with --kernel-trace only; FETCH_SIZE (KB) doubled (gfx950 tallies the 128-byte
    requests of a wide coalesced read at 64 B), WRITE_SIZE taken as is.  Returns (bytes per launch, description) or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    # not from inside a profiler run (this process is already a rocprofv3 child: the tool libraries are in its environment)
    if exe is None or os.environ.get("CV_BENCH_CHILD") or any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB",
                                                                                          "ROCPROF_OUTPUT_PATH")):
        return None
    vals = {}
    env = dict(os.environ, CV_BENCH_CHILD="1", TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cv_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--streams", "1", "--stage", "vote_decode", "--steps", "6", "--warmup", "2",
                   "--cpu-scenes", "0", "--train-steps", "0", "--measure-traffic", "0", "--points", str(a.points),
                   "--algo", str(a.algo)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if "hv_fwd_tiles" in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not v:
                return None
            vals[counter] = (sum(v) / len(v), len(v))
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f, w = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    return (2.0 * f[0] + w[0]) * 1024.0, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in two "
                                          "separate child passes of `bench.py --streams 1 --stage vote_decode` (%d / %d launches of "
                                          "hv_fwd_tiles); FETCH_SIZE %.0f KB doubled (gfx950 wide-read correction), WRITE_SIZE %.0f KB "
                                          "as is" % (f[1], w[1], f[0], w[0]))


def collective_info():
    """what aligned the timed region: backend of the process group (None for a single process without one) and whether
    librccl is mapped into this process"""
    import torch.distributed as tdist
    backend = tdist.get_backend() if tdist.is_initialized() else None
    try:
        rccl = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
    except OSError:
        rccl = []
    return {"backend": backend, "world_size": tdist.get_world_size() if tdist.is_initialized() else 1,
            "librccl_mapped": bool(rccl), "librccl": rccl[:1]}


class ResidentScene:
    """One scene with everything the timed region touches already in HBM."""

    def __init__(self, seed, n_points, dev, large=False):
        kw = dict(room=(9.0, 3.0, 9.0), n_boxes=40) if large else {}
        sc = make_scene(seed, n_points=n_points, res=RES, **kw)
        xyz, scale, prob, cls = synth_predictions(sc)
        self.host = (sc, xyz, scale, prob, cls)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.coords = t(sc.coords)
        self.coords4 = torch.cat([torch.zeros((n_points, 1), dtype=torch.int32, device=dev), self.coords], 1).contiguous()
        self.feats_in = (t(sc.feats) * 2.0 - 1.0).contiguous()       # eval_joint.py:167-168
        self.points = (self.coords * RES).float().contiguous()      # eval_joint.py:193
        self.feats = t(sc.feats)
        self.xyz, self.scale, self.prob, self.cls = t(xyz), t(scale), t(prob), t(cls)
        self.corner, _, self.dims = hv_cuda.grid_geometry(self.points, RES)
        self.v_in = hv_cuda.count_votes(self.points, self.xyz, self.scale, RES, NUM_ROTS, self.corner, self.dims)
        G = self.dims[0] * self.dims[1] * self.dims[2]
        self.cells = G
        self.vote_bytes = 40 * n_points + 192 * self.v_in + 68 * G
        self.vote_bytes_floor = 40 * n_points + 24 * G                   # compulsory traffic


KERNEL_EVENTS = True


def step_events():
    """the events of one step: [0..4] stage boundaries (recorded by run_step on the scene's stream), [5], [6] around the
    vote accumulation kernel itself (recorded by the library: cv_hv_set_kernel_events).  torch creates the hipEvent_t
    at the first record, and the library needs the handles: every event is recorded once here."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(7 if KERNEL_EVENTS else 5)]
    for e in evs:           # (all of them: cv_detect_scene_f32 records the stage boundaries too)
        e.record()
    return evs


ABLATE = ()
ADAPTIVE_SPLIT = False
SCENE_CALL = "c"
TIMED_POLICY = None     # the policy of the timed region (the parity object is computed under it)
POLICY = None           # pipeline.ScenePolicy of every run_step call (launch sizing per call: cv_scene_desc / the thread's values)


def run_step(model, hv, s, ev=None, teacher=False, keep=None):
    """One scene through eval_joint.py:163-280: network -> head -> vote -> decode -> per-class NMS.
    keep: optional dict that receives the device tensors of the step (the parity check reads them)."""
    rec = (lambda i: ev[i].record()) if ev is not None else (lambda i: None)
    if SCENE_CALL == "c" and model is not None and not ABLATE and model.USE_PROGRAM:
        # the whole scene behind ONE C call (cv_detect_scene_f32): same kernels, same order, bit-identical results; the
        # scene thread holds the GIL for one foreign call instead of ~40
        if ev is not None and len(ev) > 6:
            _lib.lib().cv_hv_set_kernel_events(ev[5].cuda_event, ev[6].cuda_event)
        try:
            dets, raw, y = pipeline.detect_scene_c(model, hv, s.coords4, s.feats_in, RES, scan_points=s.points,
                                                   predictions=(s.xyz, s.scale, s.prob, s.cls) if teacher else None,
                                                   events=ev[:5] if ev is not None else None, keep=keep,
                                                   adaptive_split=ADAPTIVE_SPLIT, policy=POLICY)
        finally:
            if ev is not None and len(ev) > 6:
                _lib.lib().cv_hv_set_kernel_events(None, None)
        return dets, raw
    with pipeline.scene_policy(POLICY):
        return _run_step_calls(model, hv, s, ev, rec, teacher, keep)


def _run_step_calls(model, hv, s, ev, rec, teacher, keep):
    """run_step, call by call (the Python pipeline: --scene-call py, the timing ablations, vote-only stages)"""
    with torch.no_grad():
        rec(0)
        hv_cuda.prefetch_geometry(s.points)       # bounds reduction of the vote grid starts before the network
        if model is not None:
            if "noplan" in ABLATE:          # timing ablation: the coordinate plan of a resident scene is built once
                x = getattr(s, "_x_cached", None)
                if x is None:
                    x = s._x_cached = ME.SparseTensor(s.feats_in, s.coords4, device=s.feats_in.device)
            else:
                x = ME.SparseTensor(s.feats_in, s.coords4, device=s.feats_in.device)   # coordinate hash + levels
            # the fp16-range flag of the network's convolutions is read after decode's wait (no extra wait per scene)
            y = model(x, defer_check=True)
            rec(1)
            net_pred = pipeline.head_joint(y.F)
            xyz, scale, prob, cls = net_pred
        else:
            rec(1)
        if model is None or teacher:
            xyz, scale, prob, cls = s.xyz, s.scale, s.prob, s.cls
        rec(2)
        if "novote" in ABLATE:
            for i in (3, 4):
                rec(i)
            return [], {}
        if ev is not None and len(ev) > 6:
            _lib.lib().cv_hv_set_kernel_events(ev[5].cuda_event, ev[6].cuda_event)
        try:
            grid_obj, grid_rot, grid_scale = hv(s.points, xyz, scale, prob)
        finally:
            if ev is not None and len(ev) > 6:
                _lib.lib().cv_hv_set_kernel_events(None, None)
        rec(3)
    if "nodecode" in ABLATE:
        rec(4)
        return [], {}
    raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, s.points, xyz, prob, cls, RES)
    rec(4)
    if model is not None and model.check_range(x, y) is not y:
        # an activation left the fp16 range: the scene is redone on the bf16 triples (never happens behind BatchNorm;
        # counted in config.range_fallbacks so that it cannot happen silently)
        with torch.no_grad():
            y = model.program_forward(x, pieces=3)
            net_pred = pipeline.head_joint(y.F)
            if not teacher:
                xyz, scale, prob, cls = net_pred
            grid_obj, grid_rot, grid_scale = hv(s.points, xyz, scale, prob)
        raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, s.points, xyz, prob, cls, RES)
    if keep is not None:
        keep.update(y=y.F if model is not None else None, net_pred=net_pred if model is not None else None,
                    grids=(grid_obj, grid_rot, grid_scale), raw=raw)
    return decode.nms_per_class(raw["boxes"], raw["scores"], raw["classes"]), raw


def cpu_oracle_scene(s, sd, full, teacher):
    """One scene through the CPU oracle; returns (stage seconds, results) - the results feed the parity flags."""
    import oracle
    from oracle import sparse_oracle as so
    sc, xyz, scale, prob, cls = s.host
    pts = sc.points
    t0 = time.perf_counter()
    y = net_pred = None
    if full:
        c4 = np.concatenate([np.zeros((len(sc.coords), 1), np.int64), sc.coords], 1)
        y = so.minkunet34c_forward(sd, c4, sc.feats * 2 - 1)
        net_pred = [a.numpy() for a in so.head_joint_eval(y)]
        if not teacher:
            xyz, scale, prob, cls = net_pred
    t1 = time.perf_counter()
    g = oracle.hv_forward(pts, xyz, scale, prob, RES, NUM_ROTS, return_vin=True)
    t2 = time.perf_counter()
    corner, _, _ = oracle.grid_geometry(pts, RES)
    d = oracle.decode(g[0], g[1], g[2], corner, RES, pts, xyz, prob, cls)
    dets = oracle.nms_per_class(d["boxes"], d["scores"], d["classes"])
    t3 = time.perf_counter()
    return dict(net=t1 - t0, vote=t2 - t1, decode=t3 - t2, total=t3 - t0), dict(y=y, net_pred=net_pred, grids=g[:3],
                                                                                v_in=int(g[3]), raw=d, dets=dets)


def parity_flags(gpu, ref, s, dec_ref):
    """Match flags of one scene, HIP path vs CPU oracle (the bar of north_star: integer outputs exact, floats 1e-4).
    Vote grids: HIP vs oracle on the same predictions.  Decode: HIP decode vs the oracle's decode of the SAME (HIP)
    grids, so a last-bit difference of an accumulated float cannot flip a tie between two stages' checks."""
    out = {"scene_seed_index": 0}
    go, gr, gs = [t.cpu().numpy() for t in gpu["grids"]]
    ro, rr, rs = ref["grids"]
    ref = dict(ref, raw=dec_ref)
    out["grid_shape_exact"] = bool(go.shape == ro.shape)
    out["v_in_exact"] = bool(s.v_in_run == ref["v_in"])
    if out["grid_shape_exact"]:
        out["touched_cells_exact"] = bool(np.array_equal(go == 0, ro == 0))
        out["grid_obj_max_rel_err"] = float(np.abs(go - ro).max() / max(1.0, float(np.abs(ro).max())))
    out["candidate_cells_exact"] = bool(np.array_equal(gpu["raw"]["cand_idx"], ref["raw"]["cand_idx"]) and
                                        np.array_equal(gpu["raw"]["verdict"], ref["raw"]["verdict"]))
    out["box_count_exact"] = bool(len(gpu["raw"]["boxes"]) == len(ref["raw"]["boxes"]))
    out["boxes"] = int(len(gpu["raw"]["boxes"]))
    out["classes_exact"] = bool(list(gpu["raw"]["classes"]) == list(ref["raw"]["classes"]))
    if out["box_count_exact"] and len(ref["raw"]["boxes"]):
        out["box_corner_max_abs_err"] = float(np.abs(gpu["raw"]["boxes"] - ref["raw"]["boxes"]).max())
    if gpu.get("y") is not None and ref.get("y") is not None:
        y, r = gpu["y"].cpu().numpy(), ref["y"].numpy()
        out["net_max_abs_err"] = float(np.abs(y - r).max())
        out["net_out_max_abs"] = float(np.abs(r).max())
        out["net_within_1e-4"] = bool(out["net_max_abs_err"] <= 1e-4 * max(1.0, out["net_out_max_abs"]))
        out["head_classes_exact"] = bool(np.array_equal(gpu["net_pred"][3].cpu().numpy(), ref["net_pred"][3]))
    return out


def cpu_baseline(a, scenes, model, hv, full, teacher):
    """CPU oracle (a port: the reference has no CPU path for the vote and its sparse engine is an absent external
    dependency) on a bounded sample: scene 0, one warm-up + best of --cpu-reps on all host threads, then ONE run with
    torch limited to 1 thread.  The same pass yields the parity flags of the HIP path on that scene."""
    import oracle
    oracle.lib()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()} if full else None
    s = scenes[0]
    nthreads = torch.get_num_threads()
    # torch-CPU thread sweep ("128 cores" ran 2.2 x slower than one thread - oversubscription of the oracle's many
    # small ops): one run per thread count after one warm-up, then --cpu-reps - 1 more at the best count; `cores` = the count
    # that gave the best time, the one-thread figure stays beside it
    sweep = sorted({t for t in ((1, 8, 16, 32, nthreads) if a.cpu_reps > 1 else (1, 16)) if t <= nthreads}) if full else [1]
    by_threads, ref = {}, None
    one = None
    try:
        torch.set_num_threads(min(8, nthreads))
        _, ref = cpu_oracle_scene(s, sd, full, teacher)          # warm-up (code paths, allocator)
        for t_n in sweep:
            torch.set_num_threads(t_n)
            t, ref = cpu_oracle_scene(s, sd, full, teacher)
            by_threads[t_n] = [t]
        best_n = min(by_threads, key=lambda n_: by_threads[n_][0]["total"])
        torch.set_num_threads(best_n)
        for _ in range(max(0, a.cpu_reps - 1)):
            t, ref = cpu_oracle_scene(s, sd, full, teacher)
            by_threads[best_n].append(t)
    finally:
        torch.set_num_threads(nthreads)
    runs = [t for v in by_threads.values() for t in v]
    best = min(by_threads[best_n], key=lambda t: t["total"])
    if full and 1 in by_threads:
        one = by_threads[1][0]

    def gpu_parity(policy):
        """scene 0 through the HIP path under `policy`, against the oracle's results of the same scene"""
        global POLICY
        before, POLICY = POLICY, policy
        try:
            keep = {}
            run_step(model, hv, s, teacher=teacher, keep=keep)
            torch.cuda.synchronize()
        finally:
            POLICY = before
        if full:
            xyz, scale = (s.xyz, s.scale) if teacher else keep["net_pred"][:2]
        else:
            xyz, scale = s.xyz, s.scale
        s.v_in_run = hv_cuda.count_votes(s.points, xyz, scale, RES, NUM_ROTS, s.corner, s.dims)
        sc, hx, hs, hp, hc = s.host
        if full and not teacher:
            hx, hs, hp, hc = [t.cpu().numpy() for t in keep["net_pred"]]
        corner, _, _ = oracle.grid_geometry(sc.points, RES)
        dec_ref = oracle.decode(*[t.cpu().numpy() for t in keep["grids"]], corner, RES, sc.points, hx, hp, hc)
        out = parity_flags(keep, ref, s, dec_ref)
        out["config"] = dict((policy or pipeline.policy_for_scenes_in_flight(1)).as_config(),
                             note="the launch sizing this scene ran under" + (" = the timed region's" if policy is TIMED_POLICY else ""))
        return out, keep

    # `parity`: under the launch sizing the timed region ran with; `parity_one_in_flight`: under the library's one-scene sizing,
    # plus whether the two runs agree bit for bit on every integer output and on the three vote grids
    par, keep_t = gpu_parity(TIMED_POLICY)
    lib_policy = pipeline.policy_for_scenes_in_flight(1)
    par_one = None
    if TIMED_POLICY is not None and TIMED_POLICY != lib_policy:
        par_one, keep_1 = gpu_parity(lib_policy)
        par_one["same_bits_as_timed_config"] = {
            "vote_grids": bool(all(torch.equal(x, y) for x, y in zip(keep_t["grids"], keep_1["grids"]))),
            "candidates_verdicts_boxes": bool(np.array_equal(keep_t["raw"]["cand_idx"], keep_1["raw"]["cand_idx"]) and
                                              np.array_equal(keep_t["raw"]["verdict"], keep_1["raw"]["verdict"]) and
                                              np.array_equal(keep_t["raw"]["boxes"], keep_1["raw"]["boxes"])),
            "net_max_abs_diff": float((keep_t["y"] - keep_1["y"]).abs().max()) if keep_t.get("y") is not None else None}
    base = {"value": 1.0 / best["total"], "unit": "scenes/s", "cores": best_n if full else 1, "kind": "port",
            "host_threads_available": nthreads,
            "thread_sweep_s": {str(n_): round(min(t["total"] for t in v), 4) for n_, v in sorted(by_threads.items())},
            "stage_s": {k: round(v, 4) for k, v in best.items()},
            "one_thread": None if one is None else {"value": 1.0 / one["total"], "cores": 1,
                                                    "stage_s": {k: round(v, 4) for k, v in one.items()}},
            "sample": "1 of the same %d-point scenes through the CPU oracle (%s): 1 warm-up, one run per torch thread count %s, "
                      "best of %d at the best count (%d threads), %.1f s of CPU work in total; build CPU oracle, not reference "
                      "code (the reference has no CPU vote and MinkowskiEngine is absent)"
                      % (a.points, "torch-CPU sparse MinkUNet34C + C vote/decode/NMS on 1 thread"
                         if full else "C vote/decode/NMS, 1 thread", sweep, len(by_threads[best_n]), best_n,
                         sum(t["total"] for t in runs))}
    return base, par, par_one


def train_batch(rank, B, n, dev, scenes=None):
    """(coords4, feats, xyz, scale, class labels) of B synthetic ScanNet-shaped scenes as train_joint.py:247-251 batches
    them (batch index in column 0, colours recentred)"""
    scenes = scenes or [make_scene(seed, n_points=n, res=RES) for seed in cvd.scene_seeds(rank, B)]
    coords = torch.cat([torch.cat([torch.full((len(s.coords), 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                        for b, s in enumerate(scenes)]).to(dev)
    feats = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(dev) * 2 - 1
    xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(dev)
    scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(dev)
    cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(dev)
    return coords, feats, xyz, scale, cls


def train_side_field(a, scenes, dev):
    """config 3 next to the headline: a few train_joint.py steps (forward + backward + Adam, fp32-level
    products, batch of --train-batch scenes) on this GPU AFTER the timed region, so that the driver's default line
    shows the training step too.  Not part of `value`."""
    from canonicalvoting_amd import train
    B = a.train_batch
    batch = train_batch(0, B, a.points, dev, scenes=[s.host[0] for s in scenes[:B]] if len(scenes) >= B else None)
    prev = ME.set_compute_dtype("fp32")
    try:
        torch.manual_seed(0)
        model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().train()
        opt = train.make_optimizer(model)
        for _ in range(2):
            train.train_step(model, opt, *batch)
        torch.cuda.synchronize()
        times = []
        for _ in range(a.train_steps):
            t0 = time.perf_counter()
            loss, _ = train.train_step(model, opt, *batch)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return {"value": float(np.median(times) * 1e3), "unit": "ms per step (median of %d)" % a.train_steps,
                "scenes_per_step": B, "points_per_scene": a.points, "dtype": "f32",
                "what": "train_joint.py:244-288 step: MinkUNet34C(3, 64) forward + backward + Adam on one GPU, "
                        "batch-statistics BatchNorm", "final_loss": float(loss)}
    finally:
        ME.set_compute_dtype(prev)
        del model, opt
        torch.cuda.empty_cache()


def main_train(a):
    """train_joint.py:244-288 steps on synthetic ScanNet-shaped batches: one process per GPU, each with its own
    batch of scenes (weak scaling), gradients all-reduced by torch DDP over RCCL, BatchNorm statistics per GPU
    (or over all ranks with --sync-bn)."""
    from canonicalvoting_amd import train
    world, rank, local = check_world(a)
    local %= torch.cuda.device_count()          # one rank per GPU on a real node; ranks share GPUs only in the launch-path test
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cvd.init("nccl", dev)
    _lib.lib()
    B, n = a.train_batch, a.points
    coords, feats, xyz, scale, cls = train_batch(rank, B, n, dev)
    ME.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
    torch.manual_seed(0)
    model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().train()
    if a.sync_bn and world > 1:
        ME.convert_sync_batchnorm(model)
    import torch.distributed as tdist
    net = train.make_ddp(model, dev) if (world > 1 or tdist.is_initialized()) else model      # (CV_DIST_FORCE=1: one-rank RCCL group)
    opt = train.make_optimizer(model)
    for _ in range(max(a.warmup, 1)):
        train.train_step(net, opt, coords, feats, xyz, scale, cls)
    cvd.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, _ = train.train_step(net, opt, coords, feats, xyz, scale, cls)
    t_enq = time.perf_counter() - t0             # the host side of the steps (launches queued, nothing waited for)
    cvd.barrier(dev)
    dt = cvd.reduce_scalar(time.perf_counter() - t0, "max", dev)
    if rank == 0:
        print(json.dumps({
            "metric": "scenes/sec (train_joint.py step: fwd + bwd + Adam, 80k-pt synthetic scans)",
            "value": a.steps * B * world / dt, "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "host_enqueue_ms_per_step": t_enq / a.steps * 1e3, "backward_overlap": ME.BACKWARD_OVERLAP,
            "config": {"workload": "train_joint.py step on %d x %d-point synthetic scenes per GPU-step, MinkUNet34C(3, 64) "
                                   "%s, Adam lr 1e-3" % (B, n, "bf16 conv products, fp32 accumulation / storage / "
                                                         "BatchNorm / optimizer" if a.dtype == "bf16" else "fp32"),
                       "parallelism": ("scene-parallel DDP x%d (RCCL gradient all-reduce, %s BatchNorm statistics)"
                                       % (world, "all-rank (SyncBN)" if a.sync_bn else "per-GPU")) if world > 1
                                      else "single GPU"},
            "final_loss": float(loss)}), flush=True)
    cvd.finalize()


def main_separate(a):
    """BASELINE config 5 as the reference runs it (eval_separate.py:136-264): NINE 8-channel per-category models on ONE
    SparseTensor per scene - the coordinate plan (sort, levels, maps, orders) is built once and shared by the nine
    forwards - then per category head split -> vote -> decode (eval_separate.py:209 slice) -> NMS.  One scene in
    flight per rank; a side mode (`--mode separate --large --points 300000`), not the headline metric.
    Vote / decode are fed per-category teacher predictions (the labels of that category), as in the default mode."""
    world, rank, local = check_world(a)
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cvd.init("nccl", dev)
    _lib.lib()
    ME.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
    ncat = 9
    torch.manual_seed(0)
    models = [MinkUNet34C(3, 8).cuda().eval() for _ in range(ncat)]          # eval_separate.py:140-150
    hv = HoughVoting(RES, NUM_ROTS)
    scenes = [ResidentScene(seed, a.points, dev, a.large) for seed in cvd.scene_seeds(rank, min(a.scenes, 2))]
    err = float(np.float32(0.3))

    def step(s, ev=None):
        rec = (lambda i: ev[i].record()) if ev is not None else (lambda i: None)
        n_det = 0
        with torch.no_grad():
            rec(0)
            x = ME.SparseTensor(s.feats_in, s.coords4, device=dev)
            x.coordinate_manager.fused_fast(5)                                  # the plan, once per scene
            rec(1)
            for c, model in enumerate(models):
                y = model(x)
                xyz_n, scale_n, prob_n = pipeline.head_separate(y.F)
                prob_c = s.prob * (s.cls == c).float()                          # teacher predictions of category c
                grid_obj, grid_rot, grid_scale = hv(s.points, s.xyz, s.scale, prob_c)
                raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, s.points, s.xyz, prob_c,
                                          torch.zeros_like(s.cls), RES, separate_variant=True, err_thresh=err)
                n_det += len(decode.nms(raw["boxes"], raw["scores"], 0.3))
                rec(2 + c)
        return n_det

    for w in range(max(a.warmup, 2)):
        step(scenes[w % len(scenes)])
    torch.cuda.synchronize()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(2 + ncat)] for _ in range(a.steps)]
    cvd.barrier(dev)
    t0 = time.perf_counter()
    n_det = 0
    for k in range(a.steps):
        n_det += step(scenes[k % len(scenes)], events[k])
    cvd.barrier(dev)
    dt = cvd.reduce_scalar(time.perf_counter() - t0, "max", dev)
    torch.cuda.synchronize()
    plan_ms = float(np.median([e[0].elapsed_time(e[1]) for e in events]))
    per_model = [float(np.median([e[1 + c].elapsed_time(e[2 + c]) for e in events])) for c in range(ncat)]
    if rank == 0:
        print(json.dumps({
            "metric": "scenes/sec (eval_separate.py path: nine per-category 8-channel models per scene)",
            "value": cvd.throughput(a.steps, world, dt), "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "single %d-point synthetic scene per GPU-step, eval_separate.py path: ONE coordinate plan, "
                                   "nine MinkUNet34C(3, 8) forwards + head + vote + decode + NMS" % a.points,
                       "points": a.points, "large": bool(a.large), "grid": scenes[0].dims, "categories": ncat,
                       "parallelism": "scene-parallel x%d, no collective" % world, "scenes_in_flight_per_gpu": 1},
            "plan_ms": plan_ms, "per_model_ms": per_model, "per_model_ms_mean": float(np.mean(per_model)),
            "plan_amortised_ms_per_model": plan_ms / ncat,
            "plan_share_if_rebuilt_per_model": plan_ms * ncat / (plan_ms * ncat + sum(per_model)),
            "detections_per_scene": n_det / a.steps}), flush=True)
    cvd.finalize()


def main_rendezvous_only(a):
    """The launch path without a GPU: what the driver's torch.distributed.run line exercises before any kernel runs."""
    world, rank, _ = check_world(a)
    cvd.init("gloo" if not torch.cuda.is_available() else "nccl", None)
    cvd.barrier(None)
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    cvd.barrier(None)
    dt = cvd.reduce_scalar(time.perf_counter() - t0, "max", None)
    if rank == 0:
        print(json.dumps({"metric": "scenes/sec (80k-pt synthetic scans)", "value": None, "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "rendezvous_only": True, "max_seconds": dt,
                          "scene_threads_per_rank": scene_threads(a.streams),
                          "scene_seeds_rank0": cvd.scene_seeds(0, a.scenes)}), flush=True)
    cvd.finalize()


def main():
    a = parse()
    if a.rendezvous_only:
        return main_rendezvous_only(a)
    if a.mode == "train":
        return main_train(a)
    if a.mode == "separate":
        return main_separate(a)
    world, rank, local = check_world(a)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    local %= torch.cuda.device_count()          # one rank per GPU on a real node; ranks share GPUs only in the launch-path test
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cvd.init("nccl", dev)
    _lib.lib()
    hv_cuda.set_algorithm(a.algo)
    ME.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
    hv = HoughVoting(RES, NUM_ROTS)
    full = a.stage == "full"
    teacher = a.predictions == "teacher" or not full
    model = None
    if full:
        torch.manual_seed(0)
        model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().eval()        # eval_joint.py:151, random init

    # scene i of rank r uses seed r*1000 + i: every rank owns different scenes
    scenes = [ResidentScene(seed, a.points, dev, a.large) for seed in cvd.scene_seeds(rank, a.scenes)]
    net_flops = None
    if full:
        with torch.no_grad():
            x0 = ME.SparseTensor(scenes[0].feats_in, scenes[0].coords4, device=dev)
            net_flops = model.forward_flops(x0)       # (pairs-based, dense-equivalent), untimed
            if not teacher:
                # the votes that are timed come from the network's predictions: count THEIR in-bounds
                # votes for the algorithmic byte count (untimed)
                for s in scenes:
                    y = model(ME.SparseTensor(s.feats_in, s.coords4, device=dev))
                    xyz, scale, prob, cls = pipeline.head_joint(y.F)
                    s.v_in = hv_cuda.count_votes(s.points, xyz, scale, RES, NUM_ROTS, s.corner, s.dims)
                    s.vote_bytes = 40 * a.points + 192 * s.v_in + 68 * s.cells
    S = scene_threads(a.streams)
    # launch sizing by the scenes the host keeps in flight (pipeline.configure_for_scenes_in_flight: what a serving host would
    # call), then the explicit overrides of the command line
    cfg = pipeline.policy_for_scenes_in_flight(S)
    split_target = a.split_target if a.split_target >= 0 else cfg.conv_split_target
    part_records = a.vote_part_records if a.vote_part_records >= 0 else cfg.vote_part_records
    masked_min_rows = a.masked_min_rows if a.masked_min_rows >= 0 else cfg.masked_min_rows
    # the policy travels with every call (cv_scene_desc.conv_split_target / vote_part_records / masked_min_rows; the thread's
    # values for the call-by-call path): nothing process-wide is touched
    global POLICY, TIMED_POLICY
    POLICY = TIMED_POLICY = pipeline.ScenePolicy(int(split_target), int(part_records), int(masked_min_rows))
    # one-call scenes size their coarse-level launches by the scenes in flight when they start (768 workgroups below four, 256
    # from four on): the tail of a short run, where the scene threads run dry one by one, gets the one-scene sizing
    global ADAPTIVE_SPLIT
    ADAPTIVE_SPLIT = bool(a.adaptive_split) and a.split_target < 0 and a.scene_call == "c"
    streams = [torch.cuda.Stream(dev) for _ in range(S)] if not a.xcd_partition else xcd_streams(dev, S, a.xcd_partition)
    # (N < 0: the last -N steps on LOW-priority streams instead - oldest scene first at the drain)
    hi_streams = [torch.cuda.Stream(dev, priority=-1 if a.tail_priority > 0 else 1) for _ in range(S)] if a.tail_priority != 0 else None
    hvs = [hv] + [HoughVoting(RES, NUM_ROTS) for _ in range(S - 1)]
    hv_cuda.reserve_pinned((8 if hi_streams else 4) * S + 8)

    import itertools
    import threading
    sys.setswitchinterval(a.switch_interval)      # GIL hand-off between the scene threads
    global KERNEL_EVENTS, ABLATE, SCENE_CALL
    SCENE_CALL = a.scene_call
    KERNEL_EVENTS = bool(a.kernel_events)
    ABLATE = tuple(x for x in a.ablate.split(",") if x)
    every = max(1, a.event_every)
    events = [step_events() if k % every == 0 else None for k in range(a.steps)]
    counts = [0] * S
    step_log = [None] * a.steps         # (thread, host time at start, at end) of every timed step
    host_us = [None] * a.steps          # cv_scene_result.host_us of every timed step
    warm_steps = [0] * S
    errors = []
    # the K steps are handed out from one counter: a thread whose scene was cheap takes the next step at once
    ticket, ticket_lock = itertools.count(), threading.Lock()
    # Warm-up runs IN the scene threads, on their streams: every (thread, stream, resident scene) pair at least twice
    # (per-stream allocator pools, per-stream scratch, pinned buffers, code objects), at least --warmup steps in total
    # and at least --min-warm-seconds of sustained work (clocks), whatever --warmup says: the driver's 20-step / 5-warm-up
    # command timed 80 ms of a chip that had seen five scenes on five of six streams.  Then every thread
    # parks on the gate BEFORE the clock starts.
    reps = max(2, -(-a.warmup // (S * len(scenes))))
    warm_t0 = time.perf_counter()
    warm_gate = threading.Barrier(S)
    gate = threading.Barrier(S + 1)

    def worker(i):
        try:
            torch.cuda.set_device(local)
            with torch.cuda.stream(streams[i]):
                for r in range(reps):
                    for j in range(len(scenes)):
                        run_step(model, hvs[i], scenes[(j + i) % len(scenes)], teacher=teacher)
                        warm_steps[i] += 1
                if hi_streams is not None:                  # the priority stream of this thread: its scratch, pools and pinned buffers
                    with torch.cuda.stream(hi_streams[i]):
                        for j in range(len(scenes)):
                            run_step(model, hvs[i], scenes[(j + i) % len(scenes)], teacher=teacher)
                            warm_steps[i] += 1
                    hi_streams[i].synchronize()
                warm_gate.wait()
                j = i
                while time.perf_counter() - warm_t0 < a.min_warm_seconds:
                    run_step(model, hvs[i], scenes[j % len(scenes)], teacher=teacher)
                    warm_steps[i] += 1
                    j += 1
                streams[i].synchronize()
                if i == 0 and "finish" in ABLATE:
                    _lib.lib().cv_sp_set_ablation(1)
                gate.wait()
                if a.stagger_us > 0 and i > 0:
                    time.sleep(i * a.stagger_us * 1e-6)
                while True:
                    with ticket_lock:
                        k = next(ticket)
                    if k >= a.steps:
                        break
                    ts = time.perf_counter()
                    if hi_streams is not None and k >= a.steps - abs(a.tail_priority):
                        with torch.cuda.stream(hi_streams[i]):
                            dets, _ = run_step(model, hvs[i], scenes[k % len(scenes)], events[k], teacher)
                    else:
                        dets, _ = run_step(model, hvs[i], scenes[k % len(scenes)], events[k], teacher)
                    step_log[k] = (i, ts, time.perf_counter())
                    if SCENE_CALL == "c" and model is not None:
                        host_us[k] = pipeline.last_scene_host_us(dev)
                    counts[i] += len(dets)
                streams[i].synchronize()
                if hi_streams is not None:
                    hi_streams[i].synchronize()
        except BaseException as e:      # a dead worker must not leave the others parked on a barrier
            errors.append(e)
            warm_gate.abort()
            gate.abort()
            raise

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(S)]
    for t in threads:
        t.start()
    while gate.n_waiting < S and not errors:       # every worker is warm and parked
        time.sleep(0.0005)
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    # no cyclic garbage collection inside the timed region: a full collection walks every object of the process with
    # the GIL held - all scene threads stand still for its milliseconds, which a 40 ms region (the driver's 20 steps)
    # shows as a 10-20 % outlier.  What the warm-up left is collected now and what survives is moved out of the
    # collector's sight (a serving process does the same after start-up).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    mem0 = torch.cuda.memory_stats(dev)
    cvd.barrier(dev)
    t0 = time.perf_counter()
    gate.wait()
    for t in threads:
        t.join()
    cvd.barrier(dev)
    dt = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    mem1 = torch.cuda.memory_stats(dev)
    # hipMalloc / hipFree calls of the caching allocator inside the timed region (0 when the warm-up did its job)
    device_allocs = {k: int(mem1.get(k, 0)) - int(mem0.get(k, 0)) for k in ("num_device_alloc", "num_device_free",
                                                                            "num_alloc_retries", "num_sync_all_streams")}
    if errors:
        raise errors[0]
    n_det = sum(counts)
    if os.environ.get("CV_BENCH_TRACE"):
        for k, (i, b, e) in enumerate(step_log):
            print("step %3d thread %d start %8.3f ms end %8.3f ms (%.3f)" % (k, i, (b - t0) * 1e3, (e - t0) * 1e3, (e - b) * 1e3),
                  file=sys.stderr)
    dt = cvd.reduce_scalar(dt, "max", dev)
    torch.cuda.synchronize()
    coll = collective_info()

    def stage_times(evs, stat=np.mean):
        return {"net": float(stat([e[0].elapsed_time(e[1]) for e in evs])),
                "head": float(stat([e[1].elapsed_time(e[2]) for e in evs])),
                "vote": float(stat([e[2].elapsed_time(e[3]) for e in evs])),
                "decode": float(stat([e[3].elapsed_time(e[4]) for e in evs]))}

    timed_steps = [k for k in range(a.steps) if events[k] is not None]
    events = [events[k] for k in timed_steps]
    vb = np.array([scenes[k % len(scenes)].vote_bytes for k in timed_steps], dtype=np.float64)
    stage_ms = stage_times(events)
    # the op (prep launches + accumulation kernel, event to event on the scene's stream) and the accumulation kernel alone
    # (events recorded by the library around hv_fwd_tiles): `roofline` prices the kernel, the op is a side field
    op_ms = np.array([e[2].elapsed_time(e[3]) for e in events])

    def kernel_times(evs, op):
        if len(evs[0]) < 7:
            return op
        k = np.array([e[5].elapsed_time(e[6]) for e in evs])
        # (the direct-atomics algorithm has no tile kernel: its events only hold the record of step_events, microseconds apart)
        return k if (k > 0.02).all() else op

    vote_ms = kernel_times(events, op_ms)
    kernel_timed = vote_ms is not op_ms
    achieved = float((vb / (vote_ms * 1e-3)).mean() / 1e9)
    op_achieved = float((vb / (op_ms * 1e-3)).mean() / 1e9)
    # Side field: the same scene threads and streams once more over a LONGER region (no events, same launch policy).  The
    # driver's 20-step region is 37 ms of a seven-deep pipeline whose scenes take 11 ms each under load - fill and drain cost
    # it ~9 % against the steady state; this region shows the steady state from inside the same process.  Never `value`.
    steady = None
    if S > 1 and world == 1 and a.steady_steps > a.steps and not ABLATE:
        n_side = int(a.steady_steps)
        ticket2, lock2, gate2, errors2 = itertools.count(), threading.Lock(), threading.Barrier(S + 1), []

        def side_worker(i):
            try:
                torch.cuda.set_device(local)
                with torch.cuda.stream(streams[i]):
                    gate2.wait()
                    if a.stagger_us > 0 and i > 0:
                        time.sleep(i * a.stagger_us * 1e-6)
                    while True:
                        with lock2:
                            k = next(ticket2)
                        if k >= n_side:
                            break
                        run_step(model, hvs[i], scenes[k % len(scenes)], None, teacher)
                    streams[i].synchronize()
            except BaseException as e:      # noqa: BLE001
                errors2.append(e)
                gate2.abort()
                raise

        side_threads = [threading.Thread(target=side_worker, args=(i,)) for i in range(S)]
        for t in side_threads:
            t.start()
        while gate2.n_waiting < S and not errors2:
            time.sleep(0.0005)
        torch.cuda.synchronize()
        gc.disable()
        t0s = time.perf_counter()
        if not errors2:
            gate2.wait()
        for t in side_threads:
            t.join()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0s
        gc.enable()
        if not errors2:
            steady = {"value": n_side / dts, "unit": "scenes/s", "steps": n_side, "ms_per_step": dts / n_side * 1e3,
                      "note": "side field, not `value`: the same threads, streams and launch policy over a longer region right after "
                              "the timed one (fill and drain of the seven-deep pipeline amortised)"}
    iso_stage = iso_achieved = iso_vote = iso_op = None
    if S > 1:
        # kernels of concurrent scenes stretch each other's event-to-event times: the same steps once more with ONE
        # scene in flight, after the timed region, give the op on its own (side fields; `frac` is the timed region's).
        # This pass runs on the main thread's stream, which has its own allocator pool and scratch: two passes over the
        # resident scenes first, then the MEDIAN over the measured steps (one cold step used to double the mean)
        iso_steps = max(min(a.steps, 48), 24)
        POLICY = pipeline.policy_for_scenes_in_flight(1)           # one scene in flight: the library's default launch sizing
        for k in range(2 * len(scenes)):
            run_step(model, hv, scenes[k % len(scenes)], teacher=teacher)
        ev2 = [step_events() for _ in range(iso_steps)]
        for k in range(iso_steps):
            run_step(model, hv, scenes[k % len(scenes)], ev2[k], teacher)
        torch.cuda.synchronize()
        iso_stage = stage_times(ev2, np.median)
        op2 = np.array([e[2].elapsed_time(e[3]) for e in ev2])
        v2 = kernel_times(ev2, op2)
        iso_vote = float(np.median(v2))
        iso_op = float(np.median(op2))
        vb2 = np.array([scenes[k % len(scenes)].vote_bytes for k in range(iso_steps)], dtype=np.float64)
        iso_achieved = float(np.median(vb2 / (v2 * 1e-3)) / 1e9)
        POLICY = TIMED_POLICY
    s0 = scenes[0]
    # HBM bytes of the vote kernel from the PMC counters are collected offline (rocprofv3 --pmc in its
    # own passes, profiles/r*/vote_hbm_traffic.json) for the default 80k workload; null otherwise
    traffic = traffic_source = None
    default_workload = a.points == N_POINTS and not a.large and a.algo in (0, 2)
    if rank == 0 and world == 1 and a.measure_traffic != 0 and default_workload:
        m = measure_vote_traffic(a)
        if m is not None:
            traffic, traffic_source = m
    for rnd in (() if traffic is not None else ("r6", "r5", "r4", "r3", "r2", "r1")):
        tj = os.path.join(ROOT, "profiles", rnd, "vote_hbm_traffic.json")
        if os.path.exists(tj) and a.points == N_POINTS and not a.large and a.algo in (0, 2):
            traffic = json.load(open(tj))["hbm_bytes_per_launch"]
            traffic_source = "file: profiles/%s/vote_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, " \
                             "profiles/vote_pmc.sh; not measured by this run)" % rnd
            break
    # counters of the regime that is timed (profiles/in_flight_counters.sh, collected offline: dispatch counters serialise the
    # kernels, so they cannot be taken inside the timed region): CU time and matrix-pipe time a scene's kernels occupy with the
    # chip to themselves, over the chip time a scene gets at the in-flight rate.  Reported as read from a file.
    in_flight = None
    ifj = os.path.join(ROOT, "profiles", "r6", "in_flight_counters.json")
    if default_workload and full and S >= 4 and os.path.exists(ifj):
        j = json.load(open(ifj))
        in_flight = {k: j[k] for k in ("cu_busy_in_flight", "mfma_busy_in_flight", "mean_kernels_running",
                                       "wall_share_with_at_least_1024_workgroups_running", "kernels_running_at_once",
                                       "fetch_mb_per_scene_raw", "write_mb_per_scene", "scenes_per_s_under_tracer") if k in j}
        in_flight["source"] = "file: profiles/r6/in_flight_counters.json - " + j.get("source", "")
    conv_peak = 2500.0 if a.dtype == "bf16" else 157.3       # dense bf16 / fp32 matrix peak, TFLOP/s
    pieces_n = 1 if a.dtype == "bf16" else 3 if (full and model.USE_PROGRAM and model.PIECES == 2) else 6
    net_ms = (iso_stage or stage_ms)["net"]
    out = {
        "metric": "scenes/sec (80k-pt synthetic scans)",
        "value": cvd.throughput(a.steps, world, dt),
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": a.dtype,
        "data": "synthetic",
        "config": {"workload": ("single %d-point synthetic scene per GPU-step, eval_joint.py path: "
                                "HIP sparse MinkUNet34C forward (%s, random init) + head + HIP vote "
                                "accumulation + decode + NMS" % (a.points, "bf16 conv products" if a.dtype == "bf16"
                                                                 else "fp32")) if full else
                               ("single %d-point synthetic scene per GPU-step: vote + decode + NMS only "
                                "(synthesised predictions)" % a.points),
                   "predictions": ("network + head run and timed; vote/decode fed with predictions synthesised from "
                                   "the labels (teacher-forced)" if full else "synthesised from labels")
                                  if teacher else "network output (random init: no cell reaches thresh_high)",
                   "points": a.points, "num_rots": NUM_ROTS, "res": RES, "grid": s0.dims,
                   "vote_algo": {0: "auto(tiles)", 1: "direct", 2: "tiles"}.get(a.algo, "ablation-%d" % a.algo),
                   "parallelism": "scene-parallel x%d, no collective" % world, "scenes_in_flight_per_gpu": S, "tail_priority_steps": a.tail_priority, "xcd_partition": a.xcd_partition,
                   "conv_split_target": "adaptive: 768 below four scenes in flight, 256 from four on" if ADAPTIVE_SPLIT else (split_target or 768),
                   "vote_part_records": part_records or 4096, "masked_min_rows": masked_min_rows,
                   **({"ablate": a.ablate, "INVALID": "timing ablation: results are wrong, not a reportable number"} if a.ablate else {})},
        "roofline": {"bound": "hbm",
                     "kernel": "hv_fwd_tiles (the accumulation kernel of cv_hv_forward_f32)" if kernel_timed
                               else "vote op (all launches of cv_hv_forward_f32)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     # what the kernel really moves through HBM per second, against the peak (the scatter itself stays in LDS)
                     "real_hbm_frac": (traffic / (float(vote_ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "isolated_real_hbm_frac": (traffic / (iso_vote * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and iso_vote) else None,
                     "avg_ms": float(vote_ms.mean()), "bytes_per_launch": float(vb.mean()),
                     "compulsory_bytes": float(s0.vote_bytes_floor), "v_in": s0.v_in,
                     "measured_in": "the timed region (HIP events %s, on the scene's stream, %d scene%s in flight)"
                                    % ("recorded by the library directly around the kernel" if kernel_timed
                                       else "around the op", S, "s" if S > 1 else ""),
                     "op_avg_ms": float(op_ms.mean()), "op_frac": op_achieved / HBM_PEAK_GBS,
                     "isolated_avg_ms": iso_vote, "isolated_achieved": iso_achieved,
                     "isolated_frac": iso_achieved / HBM_PEAK_GBS if iso_achieved else None,
                     "isolated_op_avg_ms": iso_op,
                     "note": "algorithmic-byte model of the reference's scatter (48 fp32 atomics per vote); the tile "
                             "kernel accumulates in LDS, its real HBM traffic is `traffic`"},
        "roofline_conv": None if not full else {
            "bound": "mfma", "kernel": "sparse MinkUNet34C forward (all conv launches + coordinate manager)",
            "achieved": net_flops[0] / (net_ms * 1e-3) / 1e12, "peak": conv_peak, "unit": "TFLOP/s",
            "frac": net_flops[0] / (net_ms * 1e-3) / 1e12 / conv_peak,
            "flops_per_forward": net_flops[0], "dense_equivalent_flops": net_flops[1],
            "piece_products_per_fp32_product": pieces_n if ME.CONV_X6 else None,
            "piece_flops_per_forward": pieces_n * net_flops[0] if ME.CONV_X6 else None,
            "frac_of_16bit_matrix_peak": (pieces_n * net_flops[0] / (net_ms * 1e-3) / 1e12 / 2500.0) if ME.CONV_X6 else None,
            "range_fallbacks": int(getattr(model, "range_fallbacks", 0)),
            "mfma_busy_in_flight": in_flight["mfma_busy_in_flight"] if in_flight else None,
            "measured_in": "one scene in flight" if (S == 1 or iso_stage) else "timed region",
            "note": ("opt-in bf16 compute mode: operands rounded to bf16, one product on v_mfma_f32_32x32x16_bf16, fp32 "
                     "accumulation and storage; outside the 1e-4 parity bar; " if pieces_n == 1 else
                     ("fp32 results; every fp32 product is computed as three exact fp16 x fp16 piece products "
                      "(operands split h+l: 22 significant bits and the sign of l; weights pre-scaled by a power of two) "
                      "on v_mfma_f32_32x32x16_f16 with fp32 accumulation; a convolution input beyond the fp16 range "
                      "raises a flag and the scene is redone on the bf16 triples (range_fallbacks); "
                      if pieces_n == 3 else
                      "fp32 results; every fp32 product is computed as six exact bf16 x bf16 piece products "
                      "(operands split h+m+l) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation - 0.375x the "
                      "matrix time of v_mfma_f32_32x32x2_f32 at fp32-level accuracy; ") if ME.CONV_X6 else
                     "fp32 matrix cores (v_mfma_f32_32x32x2_f32); ") +
                    "achieved counts only existing (input,output) pairs, sum 2*P*Cin*Cout over the 63 conv layers, "
                    "against the %s matrix peak" % ("bf16" if a.dtype == "bf16" else "fp32")},
        "cu_busy_in_flight": in_flight["cu_busy_in_flight"] if in_flight else None,
        "in_flight_counters": in_flight,
        "steady_state": steady,
        "detections_per_scene": n_det / a.steps,
        "collective": coll,
        "stage_ms": stage_ms,
        "stage_ms_median": stage_times(events, np.median),
        "stage_ms_isolated": iso_stage,
        "warmup_steps_run": int(sum(warm_steps)),
        # host-side latency of the timed steps (launches + the waits of one scene): a stall shows as a max far above the median
        "device_allocs_in_timed_region": device_allocs,
        "step_host_ms": {"median": float(np.median([(e - b) * 1e3 for _, b, e in step_log])),
                         "max": float(max((e - b) * 1e3 for _, b, e in step_log))},
        # where a scene call's host time goes in the timed region (cv_scene_result.host_us, medians): the coordinate plan with its
        # wait for the level counts, ENQUEUEING the network program (~100 launches, no wait), head + vote enqueue, decode + its wait
        "scene_call_host_ms": (lambda h: None if not h else dict(zip(
            ("plan_and_wait", "net_enqueue", "head_vote_enqueue", "decode_and_wait"),
            [float(np.median([x[i] for x in h])) * 1e-3 for i in range(4)])))([x for x in host_us if x is not None]),
    }
    out["cpu_baseline"] = out["parity"] = out["parity_one_in_flight"] = out["train_step_ms"] = None
    if rank == 0 and full and a.train_steps > 0 and not a.large:
        out["train_step_ms"] = train_side_field(a, scenes, dev)
    if rank == 0 and a.cpu_scenes > 0:
        out["cpu_baseline"], out["parity"], out["parity_one_in_flight"] = cpu_baseline(a, scenes, model, hv, full, teacher)
    if rank == 0:
        print(json.dumps(out), flush=True)
    cvd.finalize()


if __name__ == "__main__":
    main()
