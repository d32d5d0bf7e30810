"""Host code pinned to the REFERENCE itself (CPU tests).  tests/golden/ref_interface_*.npz were produced by running
the reference's own Interface.cc (compiled in place against include/BP_GPU.h, oracle/Makefile target `ref`,
generator tests/golden/make_ref_fixtures.py) over synthetic Pfiles: chunk plan, chunk order, every chunk's
indata / targ as the reference hands them to BP_GPU::train / CrossValid, and the weight-file bytes it writes.
This repo's reader / planner / shuffler / weight-file code (csrc/host) must reproduce the dump BYTE FOR BYTE.
When the reference tree is present (build container) the fixtures are also re-generated live and compared, and
the reference's unmodified BPtrain.cc must link against the drop-in header + libbp_hip.so with only bp_* symbols
undefined.  Nothing from /root/reference is read on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "csrc", "host")
GOLD = os.path.join(ROOT, "tests", "golden", "ref")
REF = "/root/reference"
FIXTURES = ["ref_interface_129_initwts", "ref_interface_129_randinit"]


@pytest.fixture(scope="module")
def dump_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "reader_dump")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "reader_dump.cc"),
                           os.path.join(HOST, "pfile_reader.cpp"), os.path.join(HOST, "wts_io.cpp"), "-o", exe])
    return exe


def _materialise(fx, td):
    names = dict(fea_pfile="f.pfile", targ_pfile="t.pfile", norm_file="n.norm", init_wts="mlp.0.wts")
    for k, fn in names.items():
        if k in fx:
            open(os.path.join(td, fn), "wb").write(fx[k].tobytes())
    return [str(a).replace("@DIR@", td) for a in fx["args"]]


@pytest.mark.parametrize("name", FIXTURES)
def test_host_code_reproduces_the_reference_dump(tmp_path, dump_exe, name):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    td = str(tmp_path)
    args = _materialise(fx, td)
    out = os.path.join(td, "ours.bin")
    subprocess.check_call([dump_exe, "epoch", out] + args, cwd=td)
    ours, ref = np.fromfile(out, np.uint8), fx["dump"]
    if not np.array_equal(ours, ref):                      # locate the first difference for the message
        n = min(ours.size, ref.size)
        d = np.nonzero(ours[:n] != ref[:n])[0]
        pytest.fail("%s: dump differs (sizes %d vs %d, first differing byte %s)" % (name, ours.size, ref.size, d[:1]))
    wts = np.fromfile(os.path.join(td, "mlp.1.wts"), np.uint8)
    assert np.array_equal(wts, fx["out_wts"]), "weight file bytes differ from Interface::Writeweights"
    # sanity on the fixture itself: it really walked chunks with samples
    L = int(np.frombuffer(ref[:4].tobytes(), np.int32)[0])
    assert L == 3 and ref.size > 100000


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "Interface.cc")), reason="reference tree not present (GPU box)")
def test_fixtures_are_what_the_reference_produces_today(tmp_path):
    """Build container only: compile the reference's Interface.cc in place and regenerate; the committed fixtures must
    be exactly what it emits (so they cannot drift or be hand-edited)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    for name in FIXTURES:
        fx = np.load(os.path.join(GOLD, name + ".npz"))
        td = str(tmp_path / name); os.makedirs(td)
        args = _materialise(fx, td)
        subprocess.check_call([drv, "epoch", os.path.join(td, "ref.bin")] + args, cwd=td, stdout=subprocess.DEVNULL)
        assert np.array_equal(np.fromfile(os.path.join(td, "ref.bin"), np.uint8), fx["dump"]), name
        assert np.array_equal(np.fromfile(os.path.join(td, "mlp.1.wts"), np.uint8), fx["out_wts"]), name


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "BPtrain.cc")), reason="reference tree not present (GPU box)")
def test_reference_main_links_against_the_drop_in():
    """The reference's UNMODIFIED BPtrain.cc + Interface.cc compile against include/BP_GPU.h with plain g++ and link to
    libbp_hip.so; the only undefined bp_* symbols are the six the shim calls (the drop-in boundary)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    exe = os.path.join(ROOT, "oracle", "_ref", "BPtrain_ref")
    und = subprocess.run(["nm", "-u", exe], capture_output=True, text=True, check=True).stdout
    bp = sorted(l.split()[-1] for l in und.splitlines() if " bp_" in l)
    assert bp == ["bp_create", "bp_cv_chunk", "bp_destroy", "bp_get_weights", "bp_last_error", "bp_set_hyper", "bp_train_chunk"] or \
        bp == ["bp_create", "bp_cv_chunk", "bp_destroy", "bp_get_weights", "bp_last_error", "bp_train_chunk"], bp
    assert "cuda" not in und.lower() and "cublas" not in und.lower()
