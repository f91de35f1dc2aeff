"""CPU simulation of the streaming selection rule of knn16 (appends per query under different scan orders /
threshold-refresh policies), used for DESIGN.md section 4.1: python tools/sim_appends.py [N]"""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from bench import synthetic_cells
N=int(sys.argv[1]) if len(sys.argv)>1 else 200000
X,_=synthetic_cells(N)
rng=np.random.default_rng(0)
# locality order emulation: nearest of c1 random cells, then nearest of 16 sub-centroids; greedy chains
def nearest(X, C):
    d=(X**2).sum(1)[:,None]+(C**2).sum(1)[None,:]-2*X@C.T
    return d.argmin(1)
def chain(P):
    m=len(P); D=((P[:,None,:]-P[None,:,:])**2).sum(-1); used=np.zeros(m,bool); cur=P[:,0].argmin(); used[cur]=True; rank=np.zeros(m,int)
    for s in range(1,m):
        row=np.where(used,np.inf,D[cur]); cur=row.argmin(); used[cur]=True; rank[cur]=s
    return rank
c1=int(min(64,max(8,N//4096)))
C1=X[np.sort(rng.choice(N,c1,replace=False))]
a1=nearest(X,C1); r1=chain(C1)
key=r1[a1].astype(np.int64)
leaf=np.zeros(N,np.int64)
cents=[]
for g in range(c1):
    idx=np.where(a1==g)[0]
    pick=idx[((np.arange(16)+0.5)/16*len(idx)).astype(int)]
    C2=X[pick]; a2=nearest(X[idx],C2); r2=chain(C2)
    key[idx]=key[idx]*16+r2[a2]
perm=np.argsort(key,kind='stable'); Xp=X[perm]; keyp=key[perm]
# leaves in permuted order
leaf_id=keyp; uniq,starts=np.unique(leaf_id,return_index=True); nleaf=len(uniq)
leaf_cent=np.array([Xp[starts[i]:(starts[i+1] if i+1<nleaf else N)].mean(0) for i in range(nleaf)])
print("N",N,"leaves",nleaf,"mean leaf",N/nleaf)
ksel=64; TS=64; n_tiles=(N+TS-1)//TS
Q=rng.choice(N-256,300,replace=False)
def count_appends(order_idx, d2, stale_slack=96):
    # order_idx: ref indices in scan order
    d=d2[order_idx]
    # fresh threshold
    import heapq
    fresh=0; h=[]  # max-heap of best ksel via negatives
    stale=0; thr=np.inf; buf=[]
    for v in d:
        if len(h)<ksel: heapq.heappush(h,-v); fresh+=1
        elif v< -h[0]: heapq.heapreplace(h,-v); fresh+=1
        if v<thr:
            buf.append(v); stale+=1
            if len(buf)>=ksel+stale_slack:
                buf=sorted(buf)[:ksel]; thr=buf[-1]
    return fresh,stale
res={"two_sided":[], "near_leaves":[], "random":[]}
t0=time.time()
for q in Q:
    d2=((Xp-Xp[q])**2).sum(1)
    t_own=(q//256)*4
    # two-sided tile order
    offs=[0,1,2,3]; j=0
    while len(offs)<n_tiles:
        offs.append(4+j); 
        if len(offs)<n_tiles: offs.append(-1-j)
        j+=1
    tiles=[(t_own+o)%n_tiles for o in offs]
    order=np.concatenate([np.arange(t*TS,min((t+1)*TS,N)) for t in tiles])
    res["two_sided"].append(count_appends(order,d2))
    # near-leaves first: leaves sorted by centroid distance to the query's leaf centroid, first 32 leaves, then two-sided rest
    lq=np.searchsorted(starts,q,side='right')-1
    dl=((leaf_cent-leaf_cent[lq])**2).sum(1); near=np.argsort(dl)[:32]
    seen=np.zeros(n_tiles,bool); t_list=[]
    for t in range(t_own,t_own+4):
        if not seen[t%n_tiles]: seen[t%n_tiles]=True; t_list.append(t%n_tiles)
    for l in near:
        a=starts[l]; b=starts[l+1] if l+1<nleaf else N
        for t in range(a//TS,(b-1)//TS+1):
            if not seen[t]: seen[t]=True; t_list.append(t)
    for t in tiles:
        if not seen[t]: seen[t]=True; t_list.append(t)
    order2=np.concatenate([np.arange(t*TS,min((t+1)*TS,N)) for t in t_list])
    res["near_leaves"].append(count_appends(order2,d2))
    res["random"].append(count_appends(rng.permutation(N),d2))
for k,v in res.items():
    v=np.array(v); print(k,"fresh %.0f  stale(96) %.0f"%(v[:,0].mean(),v[:,1].mean()))
print("time",time.time()-t0)
