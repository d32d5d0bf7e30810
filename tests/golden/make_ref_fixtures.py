#!/usr/bin/env python3
"""Generates tests/golden/ref_interface_*.npz by RUNNING THE REFERENCE's own host code: oracle/_ref/ref_driver
(= oracle/ref_driver.cc + /root/reference/Interface.cc compiled in place, `make -C oracle`) walks one epoch's data
path on a synthetic Pfile pair exactly as the reference's main does (BPtrain.cc:16-101) and dumps what it would
hand to the trainer.  Build container only (needs /root/reference); the fixtures are data: the input files'
bytes, the argument list and the reference's outputs (chunk plan, chunk order, every chunk's indata / targ,
weight-file bytes, log text).  fea_dim = 129 because the reference hard-codes the noise-aware block for 129 bins
(Interface.cc:776-779).

    python tests/golden/make_ref_fixtures.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pfile_util as PU  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

CASES = {
    # name: (sentence lengths, train range, cv range, traincache, targ_offset, seed, hidden, init weights file?)
    "ref_interface_129_initwts": ([40, 9, 31, 25, 18, 36, 22, 28], "0-5", "6-7", 48, 5, 345, 16, True),
    "ref_interface_129_randinit": ([23, 30, 12, 27, 11, 26], "1-4", "0-0", 1000, 0, 7, 8, False),
}


def args_for(paths, lens_case):
    lens, tr, cv, cache, toff, seed, hid, use_init = lens_case
    a = ["fea_file=" + paths["fea"], "norm_file=" + paths["norm"], "targ_file=" + paths["targ"], "outwts_file=" + paths["out"],
         "log_file=" + paths["log"], "train_sent_range=" + tr, "cv_sent_range=" + cv, "fea_dim=129", "fea_context=11",
         "targ_offset=%d" % toff, "dropoutflag=0", "traincache=%d" % cache, "bunchsize=8", "gpu_used=1",
         "init_randem_seed=%d" % seed, "momentum=0.5", "weightcost=0.0", "lrate=1", "visible_omit=0.0", "hid_omit=0.0",
         "init_randem_weight_min=-0.05", "init_randem_weight_max=0.07", "init_randem_bias_min=-0.02", "init_randem_bias_max=0.03",
         "numlayers=3", "layersizes=%d,%d,129" % (129 * 12, hid)]
    if use_init:
        a.insert(5, "initwts_file=" + paths["init"])
    return a


def make_inputs(td, case, rs):
    lens, hid, use_init = case[0], case[6], case[7]
    n = sum(lens)
    fea = (rs.normal(size=(n, 129)) * 2 + 0.5).astype(np.float32)
    tg = rs.normal(size=(n, 129)).astype(np.float32)
    mean = fea.mean(0).astype(np.float32); istd = (1.0 / fea.std(0)).astype(np.float32)
    p = {k: os.path.join(td, v) for k, v in dict(fea="f.pfile", targ="t.pfile", norm="n.norm", init="mlp.0.wts", out="mlp.1.wts",
                                                  log="mlp.1.log", dump="dump.bin").items()}
    PU.write_pfile(p["fea"], lens, fea); PU.write_pfile(p["targ"], lens, tg); PU.write_norm(p["norm"], mean, istd)
    if use_init:
        ls = [129 * 12, hid, 129]
        W = [None] + [(rs.normal(size=(ls[l - 1], ls[l])) * 0.1).astype(np.float32) for l in (1, 2)]
        b = [None] + [(rs.normal(size=ls[l]) * 0.1).astype(np.float32) for l in (1, 2)]
        PU.write_wts(p["init"], ls, W, b)
    return p


def main():
    if not os.path.exists(DRIVER):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    for name, case in CASES.items():
        rs = np.random.default_rng(len(name))
        with tempfile.TemporaryDirectory() as td:
            p = make_inputs(td, case, rs)
            args = args_for(p, case)
            subprocess.check_call([DRIVER, "epoch", p["dump"]] + args, cwd=td, stdout=subprocess.DEVNULL)
            rd = lambda k: np.frombuffer(open(p[k], "rb").read(), np.uint8)   # noqa: E731
            out = dict(fea_pfile=rd("fea"), targ_pfile=rd("targ"), norm_file=rd("norm"), dump=rd("dump"), out_wts=rd("out"),
                       log=rd("log"), args=np.array([a.replace(td, "@DIR@") for a in args]))
            if case[7]:
                out["init_wts"] = rd("init")
            np.savez_compressed(os.path.join(HERE, "ref", name + ".npz"), **out)
            print(name, {k: int(v.size) for k, v in out.items() if k != "args"})


if __name__ == "__main__":
    main()
