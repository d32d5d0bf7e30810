"""Random initial weights in the recipe the reference's tools use for ReLU nets
(toolbox/weights/gen_rand_net/Gen_rand_net.cpp:89-101 with flag=1, driven by
Gen_rand_wts_for_ReLUs_forCudaTrain.pl:7-8,14 with beta = 0.5):
W_l ~ U(-r, r), r = beta*sqrt(6)/sqrt(prev+cur); bias 0.  numpy RNG (not libc rand()), so the
values are reproducible across platforms; used by bench.py and the development tools for
synthetic runs.  Arrays are indexed like the reference's: entry 0 unused, l = 1..L-1."""
import numpy as np


def glorot_net(layersizes, seed=1, beta=0.5):
    rng = np.random.default_rng(seed)
    W, b = [None], [None]
    for l in range(1, len(layersizes)):
        p, c = layersizes[l - 1], layersizes[l]
        r = beta * np.sqrt(6.0) / np.sqrt(p + c)
        W.append(rng.uniform(-r, r, size=(p, c)).astype(np.float32))
        b.append(np.zeros(c, dtype=np.float32))
    return W, b
