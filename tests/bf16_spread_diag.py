import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')  # run from the repo root: python tests/bf16_spread_diag.py
import numpy as np
import dnnse_amd as pkg
from oracle import oracle as O, bp_numpy as N
O.build()
def rms(a,r):
    a=np.asarray(a,np.float64); r=np.asarray(r,np.float64); return float(np.sqrt(((a-r)**2).sum()/ (r**2).sum()))
for ls,B in (([300,1024,1024,1024,1024,1024,257],256), ([2827,4096,4096,4096,4096,4096,257],512)):
    W,b=N.glorot_net(ls,seed=1,beta=0.5)
    rng=np.random.default_rng(1); x=rng.standard_normal((B,ls[0]),dtype=np.float32); t=rng.standard_normal((B,ls[-1]),dtype=np.float32)
    g=pkg.BP_GPU(1,len(ls),ls,B,1.0,0.5,0.0,W,b,max_chunk_frames=B,compute_dtype=1)
    o=O.Oracle(ls,B,1.0,0.5,0.0,W,b,compute_dtype=1)
    od=O.Oracle(ls,B,1.0,0.5,0.0,W,b,compute_dtype=1,acc_double=True)
    g.train(B,x,t); o.train(x,t); od.train(x,t)
    dw,db=g.get_deltas()
    print(ls[1], "layer: gpu-vs-oracle32 | gpu-vs-oracle64acc | oracle32-vs-oracle64acc")
    for l in range(1,len(ls)):
        print("  ", l, "%.4f %.4f %.4f" % (rms(dw[l],o.dW[l]), rms(dw[l],od.dW[l]), rms(o.dW[l],od.dW[l])))
    g.close()
