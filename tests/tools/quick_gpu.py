import sys, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
import oracle
from util import *
from rlap_amd import ops
def call(ei,w,n,t,o_v,o_n,perm=None,seed=0):
    out=ops.approximate_cholesky(torch.from_numpy(ei).cuda(), None if w is None else torch.from_numpy(w).cuda(), n,t,o_v,o_n,perm=None if perm is None else torch.from_numpy(perm),seed=seed)
    return out.numpy()
u=ops.rng_uniforms(1000).cpu().numpy(); r,_=oracle.uniforms(1000); print("rng ok", np.array_equal(u,r))
for (nm,ei,n) in [("K4",clique(4),4),("K6",clique(6),6),("P9",path(9),9),("BA100_50",ba_graph(100,50,0),100),("BA500_3",ba_graph(500,3,1),500),("BA3000_10",ba_graph(3000,10,2),3000)]:
    for o_v in ["degree","random","coarsen"]:
        for o_n in ["asc","desc","random"]:
            perm=np.random.RandomState(7).permutation(n) if o_v=="random" else None
            t=n//2
            a=oracle.approximate_cholesky(ei,None,n,t,o_v,o_n,perm=perm,shuffle_seed=3)
            b=call(ei,None,n,t,o_v,o_n,perm=perm,seed=3)
            ok=a.shape==b.shape and np.array_equal(a,b)
            print(nm,o_v,o_n,a.shape,b.shape,"OK" if ok else "MISMATCH", flush=True)
            if not ok and a.shape==b.shape:
                bad=np.nonzero((a!=b).any(1))[0]; print("  first bad rows",bad[:3],a[bad[:3]],b[bad[:3]])
