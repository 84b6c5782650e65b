import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlap_amd import graphs, ops
G,n,m=64,4096,8
eis=[graphs.barabasi_albert(n,m,1000+g) for g in range(G)]
big,node_ptr=graphs.batch_disjoint(eis,[n]*G)
sc,rp=ops.approximate_cholesky_batched(big.cuda(),None,node_ptr,[n//2]*G,"degree","asc")
cnt=torch.bincount(sc[:,1].long(),minlength=G*n).cpu().numpy(); cnt=cnt[cnt>0]
print("cols",len(cnt),"rows",cnt.sum(),"max",cnt.max(),"mean",cnt.mean())
for th in (16,32,64,128,192,256,384,512,1024): print(">",th,(cnt>th).sum(), cnt[cnt>th].sum())
print(ops.last_stats)
