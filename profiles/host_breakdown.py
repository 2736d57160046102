"""Where the milliseconds of one scene's network stage go on the host side: each phase timed with a device
synchronise after it (python profiles/host_breakdown.py [points])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
sc = make_scene(3, N)
c4 = torch.cat([torch.zeros((N, 1), dtype=torch.int32), torch.from_numpy(sc.coords)], 1).to(dev)
f = (torch.from_numpy(sc.feats) * 2 - 1).to(dev)
torch.manual_seed(0)
m = MinkUNet34C(3, 64).cuda().eval()
acc = {}


def phase(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    a = acc.setdefault(name, [0.0, 0.0]); a[0] += t1 - t; a[1] += t2 - t
    return r


with torch.no_grad():
    for it in range(12):
        if it == 2:
            acc.clear()
        x = phase('SparseTensor', lambda: ME.SparseTensor(f, c4, device=dev))
        cmo = x.coordinate_manager

        def plan():
            return cmo.fused_plan()
        cm, sm, om = phase('fused_plan', plan)
        phase('maps', lambda: [cm.kernel_map(2, 1 << i, 2) for i in range(4)] + [cm.kernel_map(3, 1 << i) for i in range(5)]
              + [cm.up_map(16 >> i) for i in range(4)])
        phase('perms', lambda: [cm.mask_perms(3, 1 << i, 4) for i in range(5) if cm.num_rows(1 << i) >= 16384]
              + [cm.up_perm(16 >> i) for i in range(4)])
        phase('net', lambda: m(x))
print('%d points, ms per scene: phase  host-return  with-sync' % N)
for k, (h, s) in acc.items():
    print('%-14s %7.3f %7.3f' % (k, h / 10 * 1e3, s / 10 * 1e3))
