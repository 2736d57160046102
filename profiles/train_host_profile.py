"""host side of the training step: cProfile of train.train_step at 3 x 2000 points (the kernels are short: the step is its host
side), top functions by own time - main thread only; the autograd engine's thread is sampled through total step time"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from canonicalvoting_amd import me as ME, train
from canonicalvoting_amd.minkunet import MinkUNet34C
dev = torch.device('cuda')
batch = bench.train_batch(0, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 2000, dev)
torch.manual_seed(0)
model = MinkUNet34C(3, 64).cuda().train()
opt = train.make_optimizer(model)
for _ in range(4):
    train.train_step(model, opt, *batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    train.train_step(model, opt, *batch)
torch.cuda.synchronize()
print("step %.2f ms" % ((time.perf_counter() - t0) * 100))
# forward only / backward only host time
x = None
def fwd():
    ME._train_state.used_pairs = False
    with ME.pair_scale_hints(model):
        out = model(ME.SparseTensor(batch[1], batch[0], device=dev))
        return train.joint_loss(out.F, batch[2], batch[3], batch[4])[0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    loss = fwd()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("forward + loss, host enqueue %.2f ms" % ((t1 - t0) * 100))
with ME.pair_scale_hints(model):
    losses = []
    t_b = 0.0
    for _ in range(10):
        out = model(ME.SparseTensor(batch[1], batch[0], device=dev))
        loss = train.joint_loss(out.F, batch[2], batch[3], batch[4])[0]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss.backward()
        t_b += time.perf_counter() - t0
        torch.cuda.synchronize()
print("backward, host enqueue %.2f ms" % (t_b * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    loss = fwd()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(22)
