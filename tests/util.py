"""Shared helpers for the test-suite (golden loading, error metrics)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    ls = [int(x) for x in g["layersizes"]]
    L = len(ls)
    c = dict(name=name, ls=ls, L=L, B=int(g["B"]), steps=int(g["steps"]), act=int(g["activation"]),
             rule=int(g["momentum_rule"]), lr=float(g["lr"]), m=float(g["m"]), wc=float(g["wc"]),
             has_drop=bool(int(g["has_drop"])), drop=(float(g["drop"][0]), float(g["drop"][1])))
    c["W"] = [None] + [g["W%d" % l] for l in range(1, L)]
    c["b"] = [None] + [g["b%d" % l] for l in range(1, L)]
    c["Wf"] = [None] + [g["Wf%d" % l] for l in range(1, L)]
    c["bf"] = [None] + [g["bf%d" % l] for l in range(1, L)]
    c["gw"] = [None] + [g["gw%d" % l] if ("gw%d" % l) in g else None for l in range(1, L)]
    c["gb"] = [None] + [g["gb%d" % l] for l in range(1, L)]
    c["xs"] = [g["x%d" % s] for s in range(c["steps"])]
    c["ts"] = [g["t%d" % s] for s in range(c["steps"])]
    c["masks"] = None
    if c["has_drop"]:
        c["masks"] = [[g["mask%d_%d" % (s, l)] for l in range(L - 1)] for s in range(c["steps"])]
    c["cv_out"], c["cv_sqerr"], c["out0"] = g["cv_out"], float(g["cv_sqerr"]), g["out0"]
    return c


def relerr(a, ref):
    """max |a-ref| / max(|ref|_inf, eps): the 1e-4 contract of BASELINE.json's north_star."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


TOL = 1e-4   # north_star: "within 1e-4 relative on fp32 log-spectral frames"
