import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rlap_amd import graphs, ops
G,n,m=128,4096,8
eis=[graphs.barabasi_albert(n,m,1000+g) for g in range(G)]
big,node_ptr=graphs.batch_disjoint(eis,[n]*G)
perm=torch.from_numpy(np.concatenate([np.random.RandomState(g).permutation(n) for g in range(G)]))
sc,rp=ops.approximate_cholesky_batched(big.cuda(),None,node_ptr,[n//2]*G,"random","asc",perm=perm,seed=5)
cnt=torch.bincount(sc[:,1].long(),minlength=G*n).cpu().numpy(); cnt=cnt[cnt>0]
print("cols",len(cnt),"rows",cnt.sum(),"max",cnt.max(),"mean",cnt.mean())
for lo,hi in ((0,32),(32,64),(64,192),(192,512),(512,1024),(1024,2048),(2048,3072),(3072,3584),(3584,4096),(4096,7168),(7168,10**9)):
    sel=(cnt>lo)&(cnt<=hi); print(f"({lo},{hi}]: {sel.sum()} cols, {cnt[sel].sum()} rows ({100*cnt[sel].sum()/cnt.sum():.1f} %)")
print(dict(ops.last_stats))
