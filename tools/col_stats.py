import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlap_amd import graphs, ops
n=1_000_000
ei=graphs.barabasi_albert(n,10,2).cuda()
ops.set_timing(True)
sc=ops.approximate_cholesky(ei,None,n,n//2,"degree","asc",return_device="same")
cnt=torch.bincount(sc[:,1].long(), minlength=n).cpu().numpy()
cnt=cnt[cnt>0]
print("survivors with rows",len(cnt),"rows",cnt.sum(),"max",cnt.max())
for th in (16,32,64,128,256,512,1024,2048,4096,8192):
    print(">",th,":",(cnt>th).sum(),"cols, rows in them",cnt[cnt>th].sum())
print(ops.last_stats)
