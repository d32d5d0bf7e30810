"""The C++ drop-in header include/BP_GPU.h: a BPtrain.cc-shaped caller builds with plain g++
against it and links libbp_hip.so (CPU test); on the GPU it trains and matches the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from util import TOL, relerr

from oracle import bp_numpy as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dnn-for-speech-enhancement_amd")


def _build(tmp_path):
    import __graft_entry__
    if not os.path.exists(os.path.join(PKG, "libbp_hip.so")):
        __graft_entry__.build()
    exe = str(tmp_path / "shim_caller")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_caller.cc"), "-o", exe,
                           "-L", PKG, "-lbp_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _write_case(path, ls, B, W, b, x, t, xc, tc, hp, dropoutflag=0):
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(ls)))
        f.write(np.array(ls, np.int32).tobytes())
        f.write(struct.pack("<iiii", B, dropoutflag, x.shape[0], xc.shape[0]))
        f.write(np.array(hp, np.float32).tobytes())
        for l in range(1, len(ls)):
            f.write(np.ascontiguousarray(W[l], np.float32).tobytes())
        for l in range(1, len(ls)):
            f.write(np.ascontiguousarray(b[l], np.float32).tobytes())
        for a in (x, t, xc, tc):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())


def _case():
    ls, B = [129 * 3, 96, 129], 32
    W, b = N.glorot_net(ls, seed=21, beta=1.0)
    rng = np.random.default_rng(4)
    x = rng.normal(size=(3 * B + 5, ls[0])).astype(np.float32)
    t = rng.normal(size=(3 * B + 5, ls[-1])).astype(np.float32)
    xc = rng.normal(size=(B + 7, ls[0])).astype(np.float32)
    tc = rng.normal(size=(B + 7, ls[-1])).astype(np.float32)
    return ls, B, W, b, x, t, xc, tc, (1.0, 0.5, 0.0, 0.0, 0.0)


def test_shim_compiles_links_and_keeps_reference_error_convention(tmp_path):
    torch = pytest.importorskip("torch")
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    ls, B, W, b, x, t, xc, tc, hp = _case()
    _write_case(str(tmp_path / "in.bin"), ls, B, W, b, x, t, xc, tc, hp)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    # no device here: the reference convention is a message and exit(0) (BP_GPU.cu:20-24)
    assert r.returncode == 0
    assert "no ROCm-capable device" in r.stdout or "hipGetDeviceCount" in r.stdout
    assert not os.path.exists(tmp_path / "out.bin")


@pytest.mark.gpu
def test_shim_trains_like_the_oracle(tmp_path, oracle_mod):
    exe = _build(tmp_path)
    ls, B, W, b, x, t, xc, tc, hp = _case()
    _write_case(str(tmp_path / "in.bin"), ls, B, W, b, x, t, xc, tc, hp)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 1, r.stdout + r.stderr          # BPtrain.cc:100 "return 1" on success
    assert "this bunch has only 5 samples and is ignored." in r.stdout     # BP_GPU.cu:317
    raw = np.fromfile(str(tmp_path / "out.bin"), np.float32)
    o = oracle_mod.Oracle(ls, B, hp[0], hp[1], hp[2], W, b)
    assert o.train(x, t) == 3
    off = 1
    for l in range(1, len(ls)):
        n = ls[l - 1] * ls[l]
        assert relerr(raw[off:off + n].reshape(ls[l - 1], ls[l]), o.W[l]) < TOL
        off += n
    for l in range(1, len(ls)):
        assert relerr(raw[off:off + ls[l]], o.b[l]) < TOL
        off += ls[l]
    cv = o.crossvalid(xc, tc)
    assert abs(raw[0] - cv) < TOL * abs(cv)
