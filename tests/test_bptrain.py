"""End-to-end test of the BPtrain-compatible command line (csrc/host/bptrain.cpp) on a synthetic Pfile pair:
same `name=value` arguments as the reference's .pl driver passes, weights file and log compared with the
Python restatement of the host path (tests/pfile_util.py) + the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import pfile_util as PU
from util import TOL, relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dnn-for-speech-enhancement_amd")


def _exe():
    exe = os.path.join(PKG, "bptrain")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build()
    return exe


def test_bptrain_links_and_reports_errors_like_the_reference(tmp_path):
    """CPU: the binary exists, and a bad argument follows the reference's convention (message, exit status 0)."""
    r = subprocess.run([_exe(), "log_file"], capture_output=True, text=True)
    assert r.returncode == 0 and "Format Error" in r.stdout                      # Interface.cc:92-96
    r = subprocess.run([_exe(), "log_file=%s" % (tmp_path / "x" / "no" / "log")], capture_output=True, text=True)
    assert r.returncode == 0 and "can not open output log file" in r.stdout      # Interface.cc:246-250


@pytest.mark.gpu
@pytest.mark.parametrize("act,rule,stack", [("relu", "live", "device"), ("sigmoid", "classic", "host"),
                                            ("relu", "live", "host-noprefetch"), ("relu", "live", "device-bf16")])
def test_bptrain_epoch_matches_python_pipeline(tmp_path, oracle_mod, act, rule, stack):
    D, ctx, toff, seed, cache, B = 33, 3, 1, 345, 50, 16
    ls = [D * (ctx + 1), 64, D]                                                   # NAT block appended
    lens = [30, 22, 41, 8, 27, 35, 19, 26, 33, 24]
    rs = np.random.default_rng(9)
    n = sum(lens)
    fea = rs.normal(size=(n, D)).astype(np.float32) * 2 + 0.5
    tg = rs.normal(size=(n, D)).astype(np.float32)
    mean = fea.mean(0).astype(np.float32); istd = (1.0 / fea.std(0)).astype(np.float32)
    p = {k: str(tmp_path / v) for k, v in dict(fea="f.pfile", targ="t.pfile", norm="n.norm", init="mlp.0.wts",
                                               out="mlp.1.wts", log="mlp.1.log").items()}
    PU.write_pfile(p["fea"], lens, fea); PU.write_pfile(p["targ"], lens, tg); PU.write_norm(p["norm"], mean, istd)
    W = [None] + [(rs.normal(size=(ls[l - 1], ls[l])) * 0.1).astype(np.float32) for l in (1, 2)]
    b = [None] + [(rs.normal(size=ls[l]) * 0.1).astype(np.float32) for l in (1, 2)]
    PU.write_wts(p["init"], ls, W, b)
    args = ["fea_file=" + p["fea"], "targ_file=" + p["targ"], "norm_file=" + p["norm"], "initwts_file=" + p["init"],
            "outwts_file=" + p["out"], "log_file=" + p["log"], "train_sent_range=0-7", "cv_sent_range=8-9",
            "fea_dim=%d" % D, "fea_context=%d" % ctx, "targ_offset=%d" % toff, "dropoutflag=0", "traincache=%d" % cache,
            "bunchsize=%d" % B, "gpu_used=1", "init_randem_seed=%d" % seed, "momentum=0.5", "weightcost=0.0", "lrate=1",
            "visible_omit=0.0", "hid_omit=0.0", "numlayers=3", "layersizes=%s" % ",".join(map(str, ls)),
            "activation=" + act, "momentum_rule=" + rule]
    # frame stacking on the device (default) or on the host (the reference's Readchunk layout), with / without the
    # read-ahead thread: the same samples reach the trainer in every mode
    args += {"device": [], "host": ["stack=host"], "host-noprefetch": ["stack=host", "prefetch=0"],
             "device-bf16": ["compute=bf16"]}[stack]           # bf16 GEMM operands: oracle in the same mode, 2e-2
    bf16 = stack.endswith("bf16")
    tol = 2e-2 if bf16 else TOL
    r = subprocess.run([_exe()] + args, capture_output=True, text=True)
    assert r.returncode == 1, r.stdout + r.stderr                                  # BPtrain.cc:100
    assert "all finish!" in r.stdout
    # ---- the same epoch in Python
    mean_t = np.array([float("%.9g" % v) for v in mean], np.float32)
    istd_t = np.array([float("%.9g" % v) for v in istd], np.float32)
    fb = np.cumsum(lens).tolist(); sent_of = np.repeat(np.arange(len(lens)), lens)
    r48 = PU.Rand48(seed)                                                         # srand48 once; no random weights drawn
    starts, total = PU.plan(fb, n, ctx, cache, 0, 7)
    order_chunks = PU.rand_index(len(starts), r48)
    o = oracle_mod.Oracle(ls, B, 1.0, 0.5, 0.0, W, b, activation=1 if act == "sigmoid" else 0,
                          momentum_rule=1 if rule == "classic" else 0, compute_dtype=1 if bf16 else 0)
    log_lines = open(p["log"]).read()
    for i, ci in enumerate(order_chunks):
        cnt = total - cache * ci if ci == len(starts) - 1 else cache
        xin, xtg = PU.read_chunk(fea, tg, sent_of, fb, mean_t, istd_t, starts, total, 7, ci, ctx, cache, toff, True,
                                 PU.rand_index(cnt, r48))
        assert "Starting chunk %d of %d containing %d samples." % (i + 1, len(starts), cnt) in log_lines
        o.train(xin, xtg)
    Wg, bg = PU.read_wts(p["out"], ls)
    for l in (1, 2):
        assert relerr(Wg[l], o.W[l]) < tol and relerr(bg[l], o.b[l]) < tol
    cstarts, ctotal = PU.plan(fb, n, ctx, cache, 8, 9)
    sq = 0.0
    for ci in range(len(cstarts)):
        cnt = ctotal - cache * ci if ci == len(cstarts) - 1 else cache
        xin, xtg = PU.read_chunk(fea, tg, sent_of, fb, mean_t, istd_t, cstarts, ctotal, 9, ci, ctx, cache, toff, True,
                                 list(range(cnt)))
        sq += o.crossvalid(xin, xtg)
    m = re.search(r"CV over\. squared error: ([0-9.eE+-]+)", log_lines)             # the line the .pl greps for
    assert m and abs(float(m.group(1)) - sq / ctotal) < (tol if bf16 else 1e-3) * (sq / ctotal) + 1e-5
    for needle in ("parameters input:", "Please check...", "Norm file loaded.", "Init weight file loaded.",
                   "Get chunk info over: Training sentences have %d chunks, %d samples." % (len(starts), total),
                   "Saving over.", "Starting CV.", "Total cost time:", "Training pass: %d samples in" % total):
        assert needle in log_lines, needle


@pytest.mark.gpu
def test_bptrain_device_and_host_stacking_write_identical_weights(tmp_path):
    """stack=device (raw frames + index tables, windows built by the GPU) and stack=host (stacked rows uploaded) are
    the same computation: byte-identical weight files and CV lines, with dropout on."""
    D, ctx, cache, B = 20, 5, 64, 16
    ls = [D * (ctx + 1), 48, 32, D]
    lens = [40, 9, 33, 4, 28, 37, 21, 30]
    rs = np.random.default_rng(4)
    n = sum(lens)
    fea = rs.normal(size=(n, D)).astype(np.float32); tg = rs.normal(size=(n, D)).astype(np.float32)
    PU.write_pfile(str(tmp_path / "f"), lens, fea); PU.write_pfile(str(tmp_path / "t"), lens, tg)
    PU.write_norm(str(tmp_path / "n"), fea.mean(0).astype(np.float32), (1.0 / fea.std(0)).astype(np.float32))
    outs = {}
    for mode in ("device", "host"):
        args = ["fea_file=%s" % (tmp_path / "f"), "targ_file=%s" % (tmp_path / "t"), "norm_file=%s" % (tmp_path / "n"),
                "outwts_file=%s" % (tmp_path / ("w." + mode)), "log_file=%s" % (tmp_path / ("log." + mode)),
                "train_sent_range=0-5", "cv_sent_range=6-7", "fea_dim=%d" % D, "fea_context=%d" % ctx, "targ_offset=2",
                "dropoutflag=1", "traincache=%d" % cache, "bunchsize=%d" % B, "gpu_used=1", "init_randem_seed=7",
                "momentum=0.5", "weightcost=0.0001", "lrate=0.5", "visible_omit=0.1", "hid_omit=0.2",
                "layersizes=%s" % ",".join(map(str, ls)), "seed=99", "stack=" + mode]
        r = subprocess.run([_exe()] + args, capture_output=True, text=True)
        assert r.returncode == 1, r.stdout + r.stderr
        log = open(tmp_path / ("log." + mode)).read()
        outs[mode] = (open(tmp_path / ("w." + mode), "rb").read(), re.search(r"CV over\. squared error: (\S+)", log).group(1))
    assert outs["device"][0] == outs["host"][0] and len(outs["device"][0]) > 1000
    assert outs["device"][1] == outs["host"][1]
