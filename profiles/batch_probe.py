"""Probe: scenes per network call.  T scene threads, each runs coordinate plan + MinkUNet34C forward + head on B
scenes concatenated into one batched sparse tensor (batch index in column 0, as the training collate does).
Prints scenes/s of the network stage alone for several (T, B).  PYTHONPATH=. python profiles/batch_probe.py"""
import sys, time, threading
import numpy as np, torch
import bench
from canonicalvoting_amd import me as ME, pipeline, _lib
from canonicalvoting_amd.minkunet import MinkUNet34C

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
_lib.lib()
torch.manual_seed(0)
model = MinkUNet34C(3, 6 * 9 + 9 + 1).cuda().eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
scenes = [bench.ResidentScene(seed, N, dev) for seed in range(8)]
sys.setswitchinterval(0.0005)


def batched(B, first):
    cs, fs = [], []
    for b in range(B):
        s = scenes[(first + b) % len(scenes)]
        c = s.coords4.clone()
        c[:, 0] = b
        cs.append(c)
        fs.append(s.feats_in)
    return torch.cat(cs).contiguous(), torch.cat(fs).contiguous()


def run(T, B, seconds=0.5, split=256):
    ME.set_split_target(split)
    streams = [torch.cuda.Stream(dev) for _ in range(T)]
    inputs = [batched(B, i * B) for i in range(T)]
    gate = threading.Barrier(T + 1)
    done = [0] * T
    stop = [False]

    def worker(i):
        torch.cuda.set_device(0)
        c, f = inputs[i]
        with torch.cuda.stream(streams[i]), torch.no_grad():
            for _ in range(6):
                y = model(ME.SparseTensor(f, c, device=dev), defer_check=True)
                pipeline.head_joint(y.F)
            streams[i].synchronize()
            gate.wait()
            while not stop[0]:
                y = model(ME.SparseTensor(f, c, device=dev), defer_check=True)
                pipeline.head_joint(y.F)
                streams[i].synchronize() if T == 1 else None
                done[i] += 1
                if done[i] % 2 == 0:
                    streams[i].synchronize()
            streams[i].synchronize()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    time.sleep(seconds)
    stop[0] = True
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(done) * B
    print("threads %d  scenes/call %d  split %3d : %7.1f scenes/s  (%.3f ms/scene)" % (T, B, split, n / dt, 1e3 * dt / n), flush=True)


for T, B in ((8, 1), (1, 1), (4, 2), (8, 2), (2, 4), (4, 4), (1, 8), (2, 8), (8, 1)):
    run(T, B)
for T, B in ((4, 2), (4, 4), (2, 4)):
    run(T, B, split=0)
    run(T, B, split=512)
