"""Generates tests/golden/*.npz from the numpy fp64 restatement (oracle/bp_numpy.py).

Run here (CPU): python tests/golden/make_golden.py
The reference has no golden vectors of its own and cannot be built in this image (SURVEY.md
8c), so these fixtures pin the C oracle and the HIP path against an independent fp64
implementation of SURVEY.md Appendix A; inputs, masks and expected outputs are all stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import bp_numpy as N  # noqa: E402

CASES = [
    # name, layersizes, B, steps, activation, momentum_rule, lr, m, wc, dropout(p_vis,p_hid) or None
    ("tiny_relu", [12, 7, 5, 3], 4, 3, 0, 0, 1.0, 0.5, 0.0, None),
    ("tiny_sigmoid_classic_wc", [12, 7, 5, 3], 4, 3, 1, 1, 0.5, 0.9, 0.01, None),
    ("tiny_relu_dropout", [12, 7, 5, 3], 8, 3, 0, 0, 1.0, 0.5, 0.001, (0.25, 0.4)),
    ("nat129_small", [1548, 32, 129], 8, 2, 0, 0, 1.0, 0.5, 0.0, None),
    ("c1_sigmoid_small", [257, 64, 257], 16, 2, 1, 1, 1.0, 0.5, 0.0, None),
    ("odd_dims_relu_dropout", [70, 65, 130, 33], 12, 2, 0, 0, 1.0, 0.7, 0.0, (0.1, 0.2)),
]


def main():
    for name, ls, B, steps, act, rule, lr, m, wc, drop in CASES:
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        W, b = N.glorot_net(ls, seed=sum(map(ord, name)) + 1, beta=1.0)
        b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.1 for l in range(1, len(ls))]
        xs = [rng.normal(size=(B, ls[0])).astype(np.float32) for _ in range(steps)]
        ts = [rng.normal(size=(B, ls[-1])).astype(np.float32) for _ in range(steps)]
        masks = None
        if drop is not None:
            masks = [[(rng.random(size=(B, ls[l])) < (drop[0] if l == 0 else drop[1])).astype(np.uint8)
                      for l in range(len(ls) - 1)] for _ in range(steps)]
        Wn, bn, dWn, dbn = N.train_steps(W, b, xs, ts, lr, m, wc, masks, act, rule)
        gw, gb, ys, out0 = N.grads(W, b, xs[0], ts[0], None if masks is None else masks[0], act)
        cv_out = N.forward_cv(W, b, xs[0], 1 if drop else 0, drop[0] if drop else 0.0, drop[1] if drop else 0.0, act)
        d = dict(layersizes=np.array(ls), B=B, steps=steps, activation=act, momentum_rule=rule, lr=lr, m=m, wc=wc,
                 drop=np.array(drop if drop else (0.0, 0.0)), has_drop=int(drop is not None),
                 cv_out=cv_out, cv_sqerr=((cv_out - ts[0]) ** 2).sum(), out0=out0)
        for l in range(1, len(ls)):
            d["W%d" % l], d["b%d" % l] = W[l], b[l]
            d["Wf%d" % l], d["bf%d" % l], d["dWf%d" % l], d["dbf%d" % l] = Wn[l], bn[l], dWn[l], dbn[l]
            d["gw%d" % l], d["gb%d" % l] = gw[l], gb[l]
        for s in range(steps):
            d["x%d" % s], d["t%d" % s] = xs[s], ts[s]
            if masks is not None:
                for l in range(len(ls) - 1):
                    d["mask%d_%d" % (s, l)] = masks[s][l]
        # keep fixtures small: expected values as float64 only where tiny, else float32
        big = sum(v.size for v in d.values() if isinstance(v, np.ndarray)) > 100000
        if big:   # drop the redundant big entries, keep final weights (fp32) + outputs
            d = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 else v)
                 for k, v in d.items() if not (k.startswith("gw") or k.startswith("dWf"))}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "written")


if __name__ == "__main__":
    main()
