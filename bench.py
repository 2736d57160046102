#!/usr/bin/env python
"""Headline benchmark: scenes/sec on 80k-point synthetic scans (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one synthetic 80k-voxel scene through the hot path with its inputs already
resident in HBM.  Scenes are independent, so N ranks run N scenes per step with no data-path
collective ("scaling": "weak"); value = scenes all ranks processed / max-over-ranks time.

Scenes in flight: --streams S (default 6) host threads, each with its own HIP stream, take the K steps
from one shared counter (whichever stream is free takes the next scene), so the launch tails, small coarse-level launches and the host syncs of one scene are filled
with another scene's kernels (the per-scene work and its results are unchanged).  Kernels of concurrent
scenes stretch each other's event-to-event times, so when S > 1 the per-stage times and the roofline of
the vote op are taken in a second pass over the same K steps with ONE scene in flight, inside the same
run, after the timed region ("measured_in" says which); `value` / `ms_per_step` always come from the timed
region.  --streams 1 reproduces the one-scene-at-a-time number (profiles/r1/bench_streams1.json).

The JSON line also carries
  roofline     the vote op (zero-fill + accumulate + normalise = every launch of
               cv_hv_forward_f32) timed with HIP events on its stream, priced with the algorithmic
               bytes of DESIGN.md / SURVEY.md 8d: B_vote = 40 N + 192 V_in + 68 G.  With one scene in
               flight the events of the timed region are used; with S > 1 `achieved` / `frac` come from
               the one-scene-in-flight pass and `avg_ms_in_timed_region` / `frac_in_timed_region` give
               the same op inside the timed region, stretched by the co-running scenes' kernels
  cpu_baseline the CPU oracle (oracle/, a port - the reference has no CPU path) on a bounded
               sample of the same scenes, rank 0 only.
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
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--scenes", type=int, default=4, help="distinct resident scenes per rank")
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--algo", type=int, default=0, help="vote algorithm: 0 auto, 1 direct, 2 tiles")
    ap.add_argument("--cpu-scenes", type=int, default=1, help="scenes timed on the CPU oracle (0 = skip)")
    ap.add_argument("--stage", default="full", choices=["vote_decode", "full"],
                    help="full = MinkUNet34C forward + head + vote + decode + NMS (eval_joint.py path)")
    ap.add_argument("--streams", type=int, default=6,
                    help="scenes in flight per GPU: S host threads, each with its own HIP stream, take the steps "
                         "from one shared counter (scenes are independent; fills the launch tails and host syncs of one scene "
                         "with the kernels of another)")
    ap.add_argument("--switch-interval", type=float, default=0.0005, help="sys.setswitchinterval for the scene threads")
    ap.add_argument("--mode", default="eval", choices=["eval", "train"],
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
    ap.add_argument("--large", action="store_true",
                    help="BASELINE config 5 shaped scenes: 9x3x9 m room, 40 boxes (use with --points 300000)")
    ap.add_argument("--teacher-forced", action="store_true",
                    help="feed the vote/decode stage with predictions synthesised from the labels "
                         "(realistic peak counts) instead of the random-weight network's output")
    return ap.parse_args()


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


def run_step(model, hv, s, ev=None, teacher_forced=False):
    """One scene through eval_joint.py:163-280: network -> head -> vote -> decode -> per-class NMS."""
    rec = (lambda i: ev[i].record()) if ev is not None else (lambda i: None)
    with torch.no_grad():
        rec(0)
        hv_cuda.prefetch_geometry(s.points)       # bounds reduction of the vote grid starts before the network
        if model is not None:
            x = ME.SparseTensor(s.feats_in, s.coords4, device=s.feats_in.device)   # coordinate hash + levels
            # the fp16-range flag of the network's convolutions is read after decode's wait (no extra wait per scene)
            y = model(x, defer_check=True)
            rec(1)
            xyz, scale, prob, cls = pipeline.head_joint(y.F)
        else:
            rec(1)
        if model is None or teacher_forced:
            xyz, scale, prob, cls = s.xyz, s.scale, s.prob, s.cls
        rec(2)
        grid_obj, grid_rot, grid_scale = hv(s.points, xyz, scale, prob)
        rec(3)
    raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, s.points, xyz, prob, cls, RES)
    rec(4)
    if model is not None and model.check_range(x, y) is not y:
        # an activation left the fp16 range: the scene is redone on the bf16 triples (never happens behind BatchNorm;
        # counted in config.range_fallbacks so that it cannot happen silently)
        with torch.no_grad():
            y = model.program_forward(x, pieces=3)
            if not teacher_forced:
                xyz, scale, prob, cls = pipeline.head_joint(y.F)
            grid_obj, grid_rot, grid_scale = hv(s.points, xyz, scale, prob)
        raw = decode.decode_boxes(grid_obj, grid_rot, grid_scale, s.points, xyz, prob, cls, RES)
    return decode.nms_per_class(raw["boxes"], raw["scores"], raw["classes"]), raw


def cpu_baseline(scenes, n, model, full):
    """CPU oracle (a port: the reference has no CPU path for the vote and its sparse engine is an
    absent external dependency) on n of the same scenes, all host threads torch gives us."""
    import oracle
    from oracle import sparse_oracle as so
    oracle.lib()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()} if full else None
    t0 = time.perf_counter()
    boxes = 0
    for s in scenes[:n]:
        sc, xyz, scale, prob, cls = s.host
        pts = sc.points
        if full:
            c4 = np.concatenate([np.zeros((len(sc.coords), 1), np.int64), sc.coords], 1)
            y = so.minkunet34c_forward(sd, c4, sc.feats * 2 - 1)
            xyz, scale, prob, cls = [a.numpy() for a in so.head_joint_eval(y)]
        g = oracle.hv_forward(pts, xyz, scale, prob, RES, NUM_ROTS)
        corner, _, _ = oracle.grid_geometry(pts, RES)
        d = oracle.decode(g[0], g[1], g[2], corner, RES, pts, xyz, prob, cls)
        boxes += len(oracle.nms_per_class(d["boxes"], d["scores"], d["classes"]))
    dt = time.perf_counter() - t0
    return n / dt, dt, boxes


def main_train(a):
    """train_joint.py:244-288 steps on synthetic ScanNet-shaped batches: one process per GPU, each with its own
    batch of scenes (weak scaling), gradients all-reduced by torch DDP over RCCL, BatchNorm statistics per GPU."""
    from canonicalvoting_amd import train
    world, rank, local = cvd.world()
    local %= torch.cuda.device_count()          # one rank per GPU on a real node; ranks share GPUs only in the launch-path test
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cvd.init("nccl", dev)
    _lib.lib()
    B, n = a.train_batch, a.points
    scenes = [make_scene(seed, n_points=n, res=RES) for seed in cvd.scene_seeds(rank, B)]
    coords = torch.cat([torch.cat([torch.full((n, 1), b, dtype=torch.int32), torch.from_numpy(s.coords)], 1)
                        for b, s in enumerate(scenes)]).to(dev)
    feats = torch.cat([torch.from_numpy(s.feats) for s in scenes]).to(dev) * 2 - 1
    xyz = torch.cat([torch.from_numpy(s.xyz_labels) for s in scenes]).to(dev)
    scale = torch.cat([torch.from_numpy(s.scale_labels) for s in scenes]).to(dev)
    cls = torch.cat([torch.from_numpy(s.class_labels) for s in scenes]).to(dev)
    ME.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
    torch.manual_seed(0)
    model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().train()
    net = train.make_ddp(model, dev) if world > 1 else model
    opt = train.make_optimizer(model)
    for _ in range(max(a.warmup, 1)):
        train.train_step(net, opt, coords, feats, xyz, scale, cls)
    cvd.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, _ = train.train_step(net, opt, coords, feats, xyz, scale, cls)
    cvd.barrier(dev)
    dt = cvd.reduce_scalar(time.perf_counter() - t0, "max", dev)
    if rank == 0:
        print(json.dumps({
            "metric": "scenes/sec (train_joint.py step: fwd + bwd + Adam, 80k-pt synthetic scans)",
            "value": a.steps * B * world / dt, "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "train_joint.py step on %d x %d-point synthetic scenes per GPU-step, MinkUNet34C(3, 64) "
                                   "%s, Adam lr 1e-3" % (B, n, "bf16 conv products, fp32 accumulation / storage / "
                                                         "BatchNorm / optimizer" if a.dtype == "bf16" else "fp32"),
                       "parallelism": "scene-parallel DDP x%d (RCCL gradient all-reduce, per-GPU BatchNorm statistics)"
                                      % world if world > 1 else "single GPU"},
            "final_loss": float(loss)}), flush=True)
    cvd.finalize()


def main():
    a = parse()
    if a.mode == "train":
        return main_train(a)
    world, rank, local = cvd.world()
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
            if not a.teacher_forced:
                # the votes that are timed come from the network's predictions: count THEIR in-bounds
                # votes for the algorithmic byte count (untimed)
                for s in scenes:
                    y = model(ME.SparseTensor(s.feats_in, s.coords4, device=dev))
                    xyz, scale, prob, cls = pipeline.head_joint(y.F)
                    s.v_in = hv_cuda.count_votes(s.points, xyz, scale, RES, NUM_ROTS, s.corner, s.dims)
                    s.vote_bytes = 40 * a.points + 192 * s.v_in + 68 * s.cells
    streams = [torch.cuda.Stream(dev) for _ in range(a.streams)] if a.streams > 1 else []
    hvs = [hv] + [HoughVoting(RES, NUM_ROTS) for _ in range(a.streams - 1)]
    for w in range(a.warmup):
        if streams:
            with torch.cuda.stream(streams[w % a.streams]):
                run_step(model, hvs[w % a.streams], scenes[w % len(scenes)], teacher_forced=a.teacher_forced)
        else:
            run_step(model, hv, scenes[w % len(scenes)], teacher_forced=a.teacher_forced)
    torch.cuda.synchronize()

    events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(a.steps)]

    barrier = lambda: cvd.barrier(dev)
    barrier()
    t0 = time.perf_counter()
    n_det = 0
    if a.streams <= 1:
        for k in range(a.steps):
            dets, _ = run_step(model, hv, scenes[k % len(scenes)], events[k], a.teacher_forced)
            n_det += len(dets)
    else:
        import threading
        sys.setswitchinterval(a.switch_interval)      # GIL hand-off between the scene threads
        counts = [0] * a.streams

        # the K steps are handed out from one counter: a thread whose scene was cheap takes the next step at once
        # (a static round-robin tied thread i to scene i % 4 whenever S was a multiple of the 4 resident scenes and
        # left the threads with the lighter scenes idle at the end: 353 scenes/s at S = 4 between 401 at 3 and 404 at 6)
        import itertools
        ticket, ticket_lock = itertools.count(), threading.Lock()

        def worker(i):
            torch.cuda.set_device(local)
            with torch.cuda.stream(streams[i]):
                while True:
                    with ticket_lock:
                        k = next(ticket)
                    if k >= a.steps:
                        break
                    dets, _ = run_step(model, hvs[i], scenes[k % len(scenes)], events[k], a.teacher_forced)
                    counts[i] += len(dets)
                streams[i].synchronize()

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(a.streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        n_det = sum(counts)
    barrier()
    dt = time.perf_counter() - t0
    dt = cvd.reduce_scalar(dt, "max", dev)

    roofline_pass = "the timed region (one scene in flight)"
    vote_ms_timed = None
    if a.streams > 1:
        torch.cuda.synchronize()
        # event-to-event time of the vote op INSIDE the timed region: stretched by the kernels of the co-running scenes
        vote_ms_timed = float(np.mean([e[2].elapsed_time(e[3]) for e in events]))
        # kernels of concurrent scenes stretch each other's event-to-event times, so the per-stage times and
        # the roofline of the vote op come from a second, single-stream pass over the same steps
        torch.cuda.synchronize()
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(a.steps)]
        for k in range(a.steps):
            run_step(model, hv, scenes[k % len(scenes)], events[k], a.teacher_forced)
        torch.cuda.synchronize()
        roofline_pass = ("a second pass over the same %d steps with one scene in flight, inside this run, after "
                         "the timed region (which keeps %d scenes in flight)" % (a.steps, a.streams))
    vote_ms = np.array([e[2].elapsed_time(e[3]) for e in events])
    stage_ms = {"net": float(np.mean([e[0].elapsed_time(e[1]) for e in events])),
                "head": float(np.mean([e[1].elapsed_time(e[2]) for e in events])),
                "vote": float(vote_ms.mean()),
                "decode": float(np.mean([e[3].elapsed_time(e[4]) for e in events]))}
    vb = np.array([scenes[k % len(scenes)].vote_bytes for k in range(a.steps)], dtype=np.float64)
    achieved = float((vb / (vote_ms * 1e-3)).mean() / 1e9)
    s0 = scenes[0]
    # HBM bytes of the vote kernel from the PMC counters are collected offline (rocprofv3 --pmc in its
    # own passes, profiles/r1/vote_hbm_traffic.json) for the default 80k workload; null otherwise
    traffic = None
    tj = os.path.join(ROOT, "profiles", "r1", "vote_hbm_traffic.json")
    if os.path.exists(tj) and a.points == N_POINTS and not a.large and a.algo in (0, 2):
        traffic = json.load(open(tj))["hbm_bytes_per_launch"]
    conv_peak = 2500.0 if a.dtype == "bf16" else 157.3       # dense bf16 / fp32 matrix peak, TFLOP/s
    pieces_n = 1 if a.dtype == "bf16" else 3 if (full and model.USE_PROGRAM and model.PIECES == 2) else 6
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
                   "predictions": "synthesised from labels (teacher-forced)" if (a.teacher_forced or not full)
                                  else "network output",
                   "points": a.points, "num_rots": NUM_ROTS, "res": RES, "grid": s0.dims,
                   "vote_algo": {0: "auto(tiles)", 1: "direct", 2: "tiles"}.get(a.algo, "ablation-%d" % a.algo),
                   "parallelism": "scene-parallel x%d, no collective" % world, "scenes_in_flight_per_gpu": a.streams},
        "roofline": {"bound": "hbm", "kernel": "vote op (all launches of cv_hv_forward_f32)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_ms": float(vote_ms.mean()), "bytes_per_launch": float(vb.mean()),
                     "compulsory_bytes": float(s0.vote_bytes_floor), "v_in": s0.v_in,
                     "measured_in": roofline_pass,
                     "avg_ms_in_timed_region": vote_ms_timed,
                     "frac_in_timed_region": (float(vb.mean()) / (vote_ms_timed * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                             if vote_ms_timed else None},
        "roofline_conv": None if not full else {
            "bound": "mfma", "kernel": "sparse MinkUNet34C forward (all conv launches + coordinate manager)",
            "achieved": net_flops[0] / (stage_ms["net"] * 1e-3) / 1e12, "peak": conv_peak, "unit": "TFLOP/s",
            "frac": net_flops[0] / (stage_ms["net"] * 1e-3) / 1e12 / conv_peak,
            "flops_per_forward": net_flops[0], "dense_equivalent_flops": net_flops[1],
            "piece_products_per_fp32_product": pieces_n if ME.CONV_X6 else None,
            "piece_flops_per_forward": pieces_n * net_flops[0] if ME.CONV_X6 else None,
            "frac_of_16bit_matrix_peak": (pieces_n * net_flops[0] / (stage_ms["net"] * 1e-3) / 1e12 / 2500.0) if ME.CONV_X6 else None,
            "range_fallbacks": int(getattr(model, "range_fallbacks", 0)),
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
        "detections_per_scene": n_det / a.steps,
        "stage_ms": stage_ms,
        "stage_ms_measured_in": roofline_pass,
    }
    if rank == 0 and world == 1 and a.cpu_scenes > 0:
        nc = min(a.cpu_scenes, len(scenes))
        v, secs, _ = cpu_baseline(scenes, nc, model, full)
        out["cpu_baseline"] = {"value": v, "unit": "scenes/s", "cores": torch.get_num_threads() if full else 1,
                               "kind": "port",
                               "sample": "%d of the same %d-point scenes through the CPU oracle (%s), %.1f s; "
                                         "build CPU oracle, not reference code (the reference has no CPU "
                                         "vote and MinkowskiEngine is absent)"
                                         % (nc, a.points, "torch-CPU sparse MinkUNet34C + C vote/decode/NMS"
                                            if full else "C vote/decode/NMS, 1 thread", secs)}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    cvd.finalize()


if __name__ == "__main__":
    main()
