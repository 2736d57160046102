"""CPU statistics behind the neighbour-window design (round 5): window sizes of 128 / 256-row tiles in the spatial row order, live fraction of
the 27 offsets per 32 / 16 / 8-row block (unsorted, mask-sorted inside the tile), rows without a neighbour per mask group.
    PYTHONPATH=. python profiles/win_stats.py [points]"""
import numpy as np, sys
from canonicalvoting_amd.synth import make_scene
def spread3(v):
    v=v.astype(np.uint64)
    v=(v|(v<<np.uint64(16)))&np.uint64(0x030000FF)
    v=(v|(v<<np.uint64(8)))&np.uint64(0x0300F00F)
    v=(v|(v<<np.uint64(4)))&np.uint64(0x030C30C3)
    v=(v|(v<<np.uint64(2)))&np.uint64(0x09249249)
    return v
def order0(c):
    mn=c.min(0); ex=(c.max(0)-mn).max(); sh=0
    while (ex>>sh)>=64: sh+=1
    q=(c-mn)>>sh
    m=spread3(q[:,0])|(spread3(q[:,1])<<np.uint64(1))|(spread3(q[:,2])<<np.uint64(2))
    return np.argsort(m,kind='stable')
def nbrmap(c,ts):
    key=lambda a:(a[:,0]+40000)*10**10+(a[:,1]+40000)*10**5+(a[:,2]+40000)
    k=key(c); o=np.argsort(k); ks=k[o]
    n=len(c); M=np.full((n,27),-1,np.int64); j=0
    for dx in(-1,0,1):
        for dy in(-1,0,1):
            for dz in(-1,0,1):
                k2=key(c+np.array([dx,dy,dz])*ts)
                p=np.searchsorted(ks,k2); p[p>=n]=n-1
                hit=ks[p]==k2
                M[hit,j]=o[p[hit]]; j+=1
    return M
n=int(sys.argv[1]) if len(sys.argv)>1 else 80000
s=make_scene(0,n_points=n) if n==80000 else make_scene(0,n_points=n,room=(9.0,3.0,9.0),n_boxes=24)
c=s.coords.astype(np.int64)
c=c[order0(c)]
levels=[c]
for l in range(1,3):
    ts=1<<l
    q=(levels[-1]//ts)*ts
    _,idx=np.unique(q,axis=0,return_index=True)
    levels.append(q[np.sort(idx)])
for l,cl in enumerate(levels):
    ts=1<<l; M=nbrmap(cl,ts); n=len(cl)
    for T in (128,256):
        Ws=[];live=0;tot=0;lives=0
        for t0 in range(0,n,T):
            m=M[t0:t0+T]; v=m[m>=0]; Ws.append(len(np.unique(v)))
            for b in range(0,len(m),32):
                mb=m[b:b+32]; live+=(mb>=0).any(0).sum(); tot+=27
            # sorted in tile by mask
            keyb=((m>=0)*(1<<np.arange(27))).sum(1)
            ms=m[np.argsort(keyb,kind='stable')]
            for b in range(0,len(ms),32):
                lives+=(ms[b:b+32]>=0).any(0).sum()
        Ws=np.array(Ws)
        print(f"level {l} n={n} T={T}: W mean {Ws.mean():.0f} p50 {np.percentile(Ws,50):.0f} p99 {np.percentile(Ws,99):.0f} max {Ws.max()}  W/T {Ws.mean()/T:.2f}  live32 {live/tot:.3f} live32(sorted in tile) {lives/tot:.3f}")

print("---- liveness by block size (T=256 tiles), unsorted / sorted-in-tile by mask")
for l,cl in enumerate(levels[:2]):
    ts=1<<l; M=nbrmap(cl,ts); n=len(cl)
    for B in (32,16,8):
        live=tot=lives=lives2=0
        for t0 in range(0,n,256):
            m=M[t0:t0+256]; v=(m>=0)
            for b in range(0,len(m),B):
                live+=v[b:b+B].any(0).sum(); tot+=27
            keyb=(v*(1<<np.arange(27))).sum(1)
            vs=v[np.argsort(keyb,kind='stable')]
            for b in range(0,len(vs),B): lives+=vs[b:b+B].any(0).sum()
            # greedy: sort by gray-ish key: popcount then mask
            o=np.lexsort((keyb, v.sum(1)))
            vs=v[o]
            for b in range(0,len(vs),B): lives2+=vs[b:b+B].any(0).sum()
        print(f"level {l} B={B}: live {live/tot:.3f} sorted {lives/tot:.3f} sorted(popcount,mask) {lives2/tot:.3f}  pair density {(M>=0).mean():.3f}")

print("---- zero-mask rows per mask group (3 groups of 9 offsets, x fastest)")
for l,cl in enumerate(levels[:2]):
    ts=1<<l
    # offsets x fastest: j = (dx+1) + 3*(dy+1) + 9*(dz+1); nbrmap above loops dx outer -> recompute properly
    key=lambda a:(a[:,0]+40000)*10**10+(a[:,1]+40000)*10**5+(a[:,2]+40000)
    k=key(cl); ks=np.sort(k); n=len(cl)
    V=np.zeros((n,27),bool)
    for dz in (-1,0,1):
        for dy in (-1,0,1):
            for dx in (-1,0,1):
                j=(dx+1)+3*(dy+1)+9*(dz+1)
                V[:,j]=np.isin(key(cl+np.array([dx,dy,dz])*ts),ks)
    for g in range(3):
        m=V[:,9*g:9*g+9]
        z=(~m.any(1)).mean()
        # blocks of 32 rows in mask-sorted order that are entirely dead
        keyb=(m*(1<<np.arange(9))).sum(1)
        o=np.argsort(keyb,kind='stable')
        ms=m[o]
        dead_blocks=sum(1 for b in range(0,n,32) if not ms[b:b+32].any())
        print(f"level {l} group {g}: zero-mask rows {z:.3f}, dead 32-row blocks {dead_blocks}/{(n+31)//32}, live unit frac {np.mean([ms[b:b+32].any(0).mean() for b in range(0,n,32)]):.3f}")
