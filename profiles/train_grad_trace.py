"""activation gradients dL/d(block output) of one training step at 3 x 20k rows, HIP model vs the oracle's autograd:
walking back from the loss, where does the difference first exceed rounding?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalvoting_amd import me as ME, train
from canonicalvoting_amd.minkunet import MinkUNet34C
from canonicalvoting_amd.synth import make_scene
from oracle import sparse_oracle as so
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "20000"))
scenes = [make_scene(60 + b, n_points=N) for b in range(3)]
coords = np.concatenate([np.concatenate([np.full((N, 1), b, np.int64), s.coords], 1) for b, s in enumerate(scenes)])
feats = np.concatenate([s.feats for s in scenes]).astype(np.float32) * 2 - 1
xyz, scale, cls = [np.concatenate([getattr(s, k) for s in scenes]) for k in ("xyz_labels", "scale_labels", "class_labels")]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
torch.manual_seed(1)
model = MinkUNet34C(3, 64).to(dev).train()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
acts = {}
def hook(name):
    def f(mod, inp, out):
        out.F.retain_grad()
        acts[name] = out.F
    return f
names = ["bn0", "block1", "block2", "block3", "block4", "bntr4", "block5", "bntr5", "block6", "bntr6", "block7", "bntr7", "block8"]
for nme in names:
    getattr(model, nme).register_forward_hook(hook(nme))
out = model(ME.SparseTensor(t(feats).to(dev), t(coords).to(dev).int(), device=dev)).F
loss = train.joint_loss(out, t(xyz).to(dev), t(scale).to(dev), t(cls).to(dev))[0]
loss.backward()
# oracle with the same capture points (bn0 / bntr* hooks fire BEFORE the fused ReLU of forward_fused is applied by the
# module itself? - MinkowskiBatchNorm.forward_fused returns relu(bn(x)), the hook sees the module's forward() only when
# called through __call__; minkunet calls .forward_fused directly, so those names are skipped below when absent)
sdo = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and k.split(".")[-1] in ("kernel", "weight", "bias") else v.clone()) for k, v in sd.items()}
cm = so.CoordinateManager(coords)
oacts = {}
def keep(name, x):
    x.retain_grad(); oacts[name] = x; return x
def cbr(x, c, b, k, ts, stride=1):
    return torch.relu(so.batch_norm(so.conv(x, sdo[c + ".kernel"], cm.map(k, ts, stride)), sdo, b, True))
def up(x, c, b, tsc):
    return torch.relu(so.batch_norm(so.conv_transpose_k2s2(x, sdo[c + ".kernel"], cm.map(2, tsc // 2, 2)), sdo, b, True))
L = so.LAYERS
x = torch.from_numpy(feats)
p1 = keep("bn0", cbr(x, "conv0p1s1", "bn0", 5, 1))
o = cbr(p1, "conv1p1s2", "bn1", 2, 1, 2); b1 = keep("block1", so._layer(o, sdo, "block1", L[0], cm, 2, True))
o = cbr(b1, "conv2p2s2", "bn2", 2, 2, 2); b2 = keep("block2", so._layer(o, sdo, "block2", L[1], cm, 4, True))
o = cbr(b2, "conv3p4s2", "bn3", 2, 4, 2); b3 = keep("block3", so._layer(o, sdo, "block3", L[2], cm, 8, True))
o = cbr(b3, "conv4p8s2", "bn4", 2, 8, 2); b4 = keep("block4", so._layer(o, sdo, "block4", L[3], cm, 16, True))
o = keep("bntr4", up(b4, "convtr4p16s2", "bntr4", 16)); b5 = keep("block5", so._layer(torch.cat([o, b3], 1), sdo, "block5", L[4], cm, 8, True))
o = keep("bntr5", up(b5, "convtr5p8s2", "bntr5", 8)); b6 = keep("block6", so._layer(torch.cat([o, b2], 1), sdo, "block6", L[5], cm, 4, True))
o = keep("bntr6", up(b6, "convtr6p4s2", "bntr6", 4)); b7 = keep("block7", so._layer(torch.cat([o, b1], 1), sdo, "block7", L[6], cm, 2, True))
o = keep("bntr7", up(b7, "convtr7p2s2", "bntr7", 2)); b8 = keep("block8", so._layer(torch.cat([o, p1], 1), sdo, "block8", L[7], cm, 1, True))
yo = so.conv(b8, sdo["final.kernel"], cm.map(1, 1), sdo["final.bias"])
lo = train.joint_loss(yo, t(xyz), t(scale), t(cls))[0]
lo.backward()
print("loss", float(loss.detach()), float(lo.detach()))
for nme in reversed(names):
    if nme not in acts:
        continue
    a, b = acts[nme], oacts[nme]
    fe = float((a.detach().cpu() - b.detach()).abs().max() / b.detach().abs().max())
    ge = float((a.grad.cpu() - b.grad).abs().max() / b.grad.abs().max())
    gn = float(((a.grad.cpu() - b.grad).norm() / b.grad.norm()))
    print("%-7s rows %6d ch %3d: activation err %.2e   gradient err max %.2e  l2 %.2e" % (nme, a.shape[0], a.shape[1], fe, ge, gn))
