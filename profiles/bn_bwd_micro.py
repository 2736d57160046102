"""BatchNorm backward with and without the hl twin of dx (cv_sp_bn_backward_hl_f32): time per call, range flag, twin check.
argv: rows channels"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd import _lib, me as ME
dev = torch.device('cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
c = int(sys.argv[2]) if len(sys.argv) > 2 else 96
L = _lib.lib()
p = lambda t: t.data_ptr() if t is not None else None
x = torch.randn(n, c, device=dev); dy = torch.randn(n, c, device=dev) * 1e-4; y = torch.relu(torch.randn(n, c, device=dev))
mean, var, gamma = x.mean(0), x.var(0, unbiased=False), torch.ones(c, device=dev)
dg = torch.empty(2, c, device=dev); dx = torch.empty_like(x); dres = torch.empty_like(x); dx_hl = torch.empty_like(x)
ws = torch.empty(int(L.cv_sp_bn_workspace_bytes(c)), dtype=torch.uint8, device=dev)
slot = torch.zeros(8200, dtype=torch.int32, device=dev)
flag = ME.range_flag(dev); st = torch.cuda.current_stream().cuda_stream
def plain():
    _lib.check(L.cv_sp_bn_backward_f32(p(x), p(dy), p(y), n, c, c, p(mean), p(var), 1e-5, p(gamma), p(dg[0]), p(dg[1]), p(dx), p(dres), p(ws), ws.numel(), st), "bn")
def twin():
    _lib.check(L.cv_sp_bn_backward_hl_f32(p(x), p(dy), p(y), n, c, c, p(mean), p(var), 1e-5, p(gamma), p(dg[0]), p(dg[1]), p(dx), p(dres), p(ws), ws.numel(), p(dx_hl), p(slot), flag.data_ptr(), None, st), "bn hl")
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("plain %.1f us" % timed(plain))
print("twin, no maximum yet (s = 1) %.1f us, flag %d" % (timed(twin), int(flag[0])))
slot[4096:8192] = slot[0:4096]; slot[0:4096] = 0
print("twin, scaled %.1f us, flag %d, 1/s %g, max |dx| %g" % (timed(twin), int(flag[0]), float(slot[8192:8193].view(torch.float32)), float(dx.abs().max())))
inv = float(slot[8192:8193].view(torch.float32))
back = ME.from_hl(dx_hl) * inv
print("twin * 1/s against dx: max |d| / max |dx| = %.2e" % (float((back - dx).abs().max()) / float(dx.abs().max())))
