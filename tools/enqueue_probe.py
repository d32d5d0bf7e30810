import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, dnnse_amd
LAYERS=[2827,2048,2048,2048,257]; B=256; chunk=102400
W,b=dnnse_amd.glorot_net(LAYERS,seed=1,beta=0.5)
g=dnnse_amd.BP_GPU(1,5,LAYERS,B,1.0,0.5,0.0,W,b,dropoutflag=1,visible_omit=0.1,hid_omit=0.2,seed=1,max_chunk_frames=chunk)
g.fill_chunk_synthetic(chunk,1); g.train_resident(0,40*B); g.sync()
for rep in range(3):
    t0=time.perf_counter(); g.train_resident(0,400*B); t1=time.perf_counter(); g.sync(); t2=time.perf_counter()
    print("enqueue %.3f ms/step, total %.3f ms/step" % ((t1-t0)/400*1e3, (t2-t0)/400*1e3))
