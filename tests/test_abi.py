"""CPU tests: the C-ABI library loads and exports every symbol include/bp_c_api.h declares
(no compute without a GPU), and the host mirror fails loudly instead of falling back."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "bp_c_api.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = pkg.load_library()
    decl = _declared_symbols()
    assert len(decl) >= 15
    for s in decl:
        assert hasattr(lib, s), "missing export %s" % s
    assert sorted(pkg.ABI_SYMBOLS) == decl
    assert lib.bp_build_target() == b"gfx950"
    assert lib.bp_abi_version() == 5


def test_config_struct_matches_header(pkg):
    hdr = open(os.path.join(ROOT, "include", "bp_c_api.h")).read()
    body = re.search(r"typedef struct bp_config \{(.*?)\} bp_config;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(int|float|uint64_t)\s+", "", decl)
        for n in decl.split(","):
            names.append(re.sub(r"\[.*\]", "", n).strip())
    assert names == [f[0] for f in pkg.BPConfig._fields_]


def test_no_cpu_fallback_without_gpu(pkg):
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    w = [None, np.zeros((4, 3), np.float32), np.zeros((3, 2), np.float32)]
    b = [None, np.zeros(3, np.float32), np.zeros(2, np.float32)]
    with pytest.raises(pkg.BPError):
        pkg.BP_GPU(1, 3, [4, 3, 2], 4, 1.0, 0.5, 0.0, w, b)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in ("dnn-for-speech-enhancement_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".cc")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"(import\s+oracle|from\s+oracle|oracle/|bp_oracle|libbp_oracle)", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
