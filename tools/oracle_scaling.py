"""Development helper: how the CPU oracle's C2 step scales with OMP threads on this host (run once per thread count)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O, bp_numpy as N
ls = [2827, 2048, 2048, 2048, 257]
W, b = N.glorot_net(ls, seed=1, beta=0.5)
rng = np.random.default_rng(1)
x = rng.standard_normal((256, ls[0]), dtype=np.float32); t = rng.standard_normal((256, 257), dtype=np.float32)
for drop in (0, 1):
    o = O.Oracle(ls, 256, 1.0, 0.5, 0.0, W, b, dropoutflag=drop, visible_omit=0.1, hid_omit=0.2, seed=1)
    o.forward(x)
    t0 = time.perf_counter(); o.forward(x); tf = time.perf_counter() - t0
    o.train_bunch(x, t)
    t0 = time.perf_counter(); o.train_bunch(x, t); ts = time.perf_counter() - t0
    print("threads %s dropout %d: forward %.3f s, train step %.3f s (%.0f frames/s)" % (os.environ.get("OMP_NUM_THREADS", "all"), drop, tf, ts, 256 / ts))
