"""The drop-in claim tested with the REFERENCE's OWN main (-m gpu): oracle/_ref/BPtrain_ref is the reference's
unmodified BPtrain.cc + Interface.cc compiled in the build container against include/BP_GPU.h and linked to
libbp_hip.so (oracle/Makefile, target `ref`; the binary travels to the GPU box, the sources do not).  On the same
synthetic Pfile pair it must produce the SAME weight-file bytes and the same log lines as this repo's `bptrain`
(csrc/host/bptrain.cpp): both drive the same library with -- if the host code is a faithful restatement --
bit-identical inputs in the same order.  Also: bptrain gpu_used=N really trains data-parallel on N ranks."""
import os
import re
import subprocess

import numpy as np
import pytest

import pfile_util as PU
from util import TOL, relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "BPtrain_ref")
OUR_EXE = os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "bptrain")
GOLD = os.path.join(ROOT, "tests", "golden", "ref")


def _inputs(td, name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    for k, fn in dict(fea_pfile="f.pfile", targ_pfile="t.pfile", norm_file="n.norm", init_wts="mlp.0.wts").items():
        if k in fx:
            open(os.path.join(td, fn), "wb").write(fx[k].tobytes())
    return [str(a).replace("@DIR@", td) for a in fx["args"]]


def _set(args, **kv):
    out = [a for a in args if a.split("=")[0] not in kv]
    return out + ["%s=%s" % (k, v) for k, v in kv.items()]


@pytest.mark.skipif(not os.path.exists(REF_EXE), reason="oracle/_ref/BPtrain_ref not built (needs the reference tree at build time)")
@pytest.mark.parametrize("name,drop", [("ref_interface_129_initwts", False), ("ref_interface_129_randinit", True)])
def test_reference_main_and_bptrain_write_identical_files(tmp_path, name, drop):
    outs = {}
    for who, exe in (("ref", REF_EXE), ("ours", OUR_EXE)):
        td = str(tmp_path / who); os.makedirs(td)
        args = _set(_inputs(td, name), lrate="0.05")          # (the fixture's lrate=1 diverges on this random net)
        env = dict(os.environ)
        if drop:                                    # dropout on: same Philox key for both (the reference seeds from time())
            args = _set(args, dropoutflag=1, visible_omit=0.1, hid_omit=0.2)
            env["BP_SEED"] = "4242"
            if who == "ours":
                args = args + ["seed=4242"]
        r = subprocess.run([exe] + args, cwd=td, env=env, capture_output=True, text=True)
        assert r.returncode == 1, (who, r.stdout[-2000:], r.stderr[-2000:])                # BPtrain.cc:100
        assert "all finish!" in r.stdout
        log = open(os.path.join(td, "mlp.1.log")).read().replace(td, "@")
        outs[who] = (open(os.path.join(td, "mlp.1.wts"), "rb").read(), log)
    assert outs["ref"][0] == outs["ours"][0] and len(outs["ref"][0]) > 10000, "weight files differ"
    # every line the reference logs appears in ours verbatim, except its wall-clock line
    ours_lines = set(outs["ours"][1].splitlines())
    for line in outs["ref"][1].splitlines():
        if line.startswith("Total cost time"):
            continue
        assert line in ours_lines, "reference log line missing from bptrain's log: %r" % line
    cv = [re.search(r"CV over\. squared error: (\S+)", o[1]).group(1) for o in (outs["ref"], outs["ours"])]
    assert cv[0] == cv[1] and np.isfinite(float(cv[0]))


def test_bptrain_gpu_used_2_trains_data_parallel(tmp_path, oracle_mod):
    """gpu_used=2: two forked ranks (sharing device 0 here), bunchsize = global minibatch, library-internal exchange.
    The weights file must equal the oracle trained on the global minibatch (1e-4) and be close to gpu_used=1."""
    D, ctx, toff, seed, cache, B = 33, 3, 1, 345, 64, 16
    ls = [D * (ctx + 1), 64, D]
    lens = [30, 22, 41, 8, 27, 35, 19, 26, 33, 24]
    rs = np.random.default_rng(9)
    n = sum(lens)
    fea = rs.normal(size=(n, D)).astype(np.float32) * 2 + 0.5
    tg = rs.normal(size=(n, D)).astype(np.float32)
    mean = fea.mean(0).astype(np.float32); istd = (1.0 / fea.std(0)).astype(np.float32)
    W = [None] + [(rs.normal(size=(ls[l - 1], ls[l])) * 0.1).astype(np.float32) for l in (1, 2)]
    b = [None] + [(rs.normal(size=ls[l]) * 0.1).astype(np.float32) for l in (1, 2)]
    res = {}
    for gpus, stack in ((1, "device"), (2, "device"), (2, "host")):
        td = str(tmp_path / ("g%d%s" % (gpus, stack))); os.makedirs(td)
        p = {k: os.path.join(td, v) for k, v in dict(fea="f.pfile", targ="t.pfile", norm="n.norm", init="mlp.0.wts",
                                                      out="mlp.1.wts", log="mlp.1.log").items()}
        PU.write_pfile(p["fea"], lens, fea); PU.write_pfile(p["targ"], lens, tg); PU.write_norm(p["norm"], mean, istd)
        PU.write_wts(p["init"], ls, W, b)
        args = ["fea_file=" + p["fea"], "targ_file=" + p["targ"], "norm_file=" + p["norm"], "initwts_file=" + p["init"],
                "outwts_file=" + p["out"], "log_file=" + p["log"], "train_sent_range=0-7", "cv_sent_range=8-9",
                "fea_dim=%d" % D, "fea_context=%d" % ctx, "targ_offset=%d" % toff, "dropoutflag=1", "traincache=%d" % cache,
                "bunchsize=%d" % B, "gpu_used=%d" % gpus, "init_randem_seed=%d" % seed, "momentum=0.5", "weightcost=0.0",
                "lrate=1", "visible_omit=0.1", "hid_omit=0.2", "layersizes=%s" % ",".join(map(str, ls)), "seed=77",
                "stack=" + stack, "device=0" if gpus == 1 else "prefetch=1"]
        env = dict(os.environ, BP_DP_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([OUR_EXE] + args, cwd=td, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "all finish!" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        log = open(p["log"]).read()
        res[(gpus, stack)] = (PU.read_wts(p["out"], ls), float(re.search(r"CV over\. squared error: (\S+)", log).group(1)),
                              open(p["out"], "rb").read())
    assert res[(2, "device")][2] == res[(2, "host")][2]                 # stacking mode does not change a bit, also under DP
    # oracle on the global minibatch, same Philox key (masks are keyed by the global frame index)
    mean_t = np.array([float("%.9g" % v) for v in mean], np.float32)
    istd_t = np.array([float("%.9g" % v) for v in istd], np.float32)
    fb = np.cumsum(lens).tolist(); sent_of = np.repeat(np.arange(len(lens)), lens)
    r48 = PU.Rand48(seed)
    starts, total = PU.plan(fb, n, ctx, cache, 0, 7)
    o = oracle_mod.Oracle(ls, B, 1.0, 0.5, 0.0, W, b, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=77)
    for ci in PU.rand_index(len(starts), r48):
        cnt = total - cache * ci if ci == len(starts) - 1 else cache
        xin, xtg = PU.read_chunk(fea, tg, sent_of, fb, mean_t, istd_t, starts, total, 7, ci, ctx, cache, toff, True,
                                 PU.rand_index(cnt, r48))
        o.train(xin, xtg)
    for key in ((1, "device"), (2, "device")):
        (Wg, bg), cv, _ = res[key]
        for l in (1, 2):
            assert relerr(Wg[l], o.W[l]) < TOL and relerr(bg[l], o.b[l]) < TOL, (key, l)
    assert abs(res[(1, "device")][1] - res[(2, "device")][1]) < 1e-3 * abs(res[(1, "device")][1])
