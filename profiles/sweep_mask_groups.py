import sys, json, subprocess
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, time
from canonicalvoting_amd.minkunet import MinkUNet34C, MinkUNetBase
from canonicalvoting_amd import me as ME
from canonicalvoting_amd.synth import make_scene
dev=torch.device('cuda')
sc=make_scene(3,80000)
c4=torch.cat([torch.zeros((80000,1),dtype=torch.int32),torch.from_numpy(sc.coords)],1).to(dev)
f=(torch.from_numpy(sc.feats)*2-1).to(dev)
torch.manual_seed(0)
m=MinkUNet34C(3,64).cuda().eval()
def run(n=8):
    with torch.no_grad():
        for _ in range(3): m(ME.SparseTensor(f,c4,device=dev))
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(n): y=m(ME.SparseTensor(f,c4,device=dev))
        torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
for groups,minrows in ((3,16384),(3,4096),(3,2048),(4,16384),(3,10**9)):
    MinkUNetBase.MASK_GROUPS=groups; MinkUNetBase.MASKED_MIN_ROWS=minrows
    print('groups',groups,'minrows',minrows,'net ms %.3f'%run())
