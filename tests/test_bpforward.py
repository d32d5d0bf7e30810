"""bpforward (csrc/host/bpforward.cpp, "next" row N4): noisy-LPS Pfile in -> enhanced-LPS Pfile out on the MI355X, checked
against the oracle's CV forward (cv_bunch_single semantics, BP_GPU.cu:676-773) on the Python restatement of the reader."""
import os
import struct
import subprocess

import numpy as np
import pytest

import pfile_util as PU
from util import TOL, relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "bpforward")


def read_pfile(path, dim):
    raw = open(path, "rb").read()
    hdr = raw[:32768].split(b"\0")[0].decode()
    ns = int(hdr.split("-num_sentences")[1].split()[0]); nf = int(hdr.split("-num_frames")[1].split()[0])
    rec = np.frombuffer(raw, ">u4", nf * (2 + dim), 32768).reshape(nf, 2 + dim)
    table = np.frombuffer(raw, ">i4", ns + 1, 32768 + nf * (2 + dim) * 4)
    return rec[:, 0].astype(int), rec[:, 1].astype(int), rec[:, 2:].astype("<u4").view("<f4"), table


def test_bpforward_reports_errors_like_the_reference():
    r = subprocess.run([EXE, "fea_file"], capture_output=True, text=True)
    assert r.returncode == 0 and "Format Error" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("drop", [False, True])
def test_bpforward_matches_oracle_forward(tmp_path, oracle_mod, drop):
    D, ctx, toff = 33, 5, 2
    ls = [D * (ctx + 1), 96, 64, D]
    lens = [30, 3, 41, 27, 12]
    rs = np.random.default_rng(2)
    n = sum(lens)
    fea = rs.normal(size=(n, D)).astype(np.float32) * 2 + 0.5
    mean = fea.mean(0).astype(np.float32); istd = (1.0 / fea.std(0)).astype(np.float32)
    p = {k: str(tmp_path / v) for k, v in dict(fea="f.pfile", norm="n.norm", wts="mlp.wts", out="enh.pfile").items()}
    PU.write_pfile(p["fea"], lens, fea); PU.write_norm(p["norm"], mean, istd)
    W = [None] + [(rs.normal(size=(ls[l - 1], ls[l])) * 0.1).astype(np.float32) for l in range(1, 4)]
    b = [None] + [(rs.normal(size=ls[l]) * 0.1).astype(np.float32) for l in range(1, 4)]
    PU.write_wts(p["wts"], ls, W, b)
    args = ["fea_file=" + p["fea"], "norm_file=" + p["norm"], "initwts_file=" + p["wts"], "out_file=" + p["out"],
            "layersizes=%s" % ",".join(map(str, ls)), "fea_dim=%d" % D, "fea_context=%d" % ctx, "targ_offset=%d" % toff,
            "traincache=20", "bunchsize=16"] + (["dropoutflag=1", "visible_omit=0.1", "hid_omit=0.2"] if drop else [])
    r = subprocess.run([EXE] + args, capture_output=True, text=True)
    assert r.returncode == 1 and "enhanced" in r.stdout, r.stdout + r.stderr
    sid, fid, feats, table = read_pfile(p["out"], D)
    # expected: every window of every sentence long enough, in file order
    mean_t = np.array([float("%.9g" % v) for v in mean], np.float32)
    istd_t = np.array([float("%.9g" % v) for v in istd], np.float32)
    x = PU.expected_windows(fea, lens, mean_t, istd_t, ctx, True)     # EVERY window of every sentence, file order
    total = x.shape[0]
    kw = dict(dropoutflag=1, visible_omit=0.1, hid_omit=0.2) if drop else {}
    o = oracle_mod.Oracle(ls, 16, weights=W, bias=b, **kw)
    assert feats.shape == (x.shape[0], D) and relerr(feats, o.forward(x)) < TOL
    exp_sid = np.concatenate([np.full(max(0, ln - ctx + 1), s) for s, ln in enumerate(lens)])
    exp_fid = np.concatenate([np.arange(max(0, ln - ctx + 1)) + toff for ln in lens])
    # every (sentence, frame) exactly once and in order: the enhancement chunking cuts on sentence boundaries and splits
    # over-long sentences with an overlap (traincache=20 < the 26 / 37 / 23 windows of three of the sentences here)
    assert len(sid) == total and sid.tolist() == exp_sid.tolist() and fid.tolist() == exp_fid.tolist()
    assert table[0] == 0 and table[-1] == total and len(table) == len(lens) + 1
