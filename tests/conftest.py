import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Measured parity numbers of a run (flip counts, per-tensor errors, which bar applied): the tests hand them to the
# `parity_record` fixture instead of only print()ing them, and the session leaves ONE JSON behind --
# $BP_PARITY_JSON if set, else gpurun_out/parity_numbers.json under the repo root (the directory gpurun merges back).
# The builder-side copy of a round is committed as profiles/rNN_parity_numbers.json.
_PARITY = {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _jsonable(v):
    import numpy as np
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return v.item()
    if isinstance(v, np.ndarray):
        return v.tolist()
    return v


@pytest.fixture
def parity_record(request):
    """record(key=value, ...): merged into this test's entry of the session's parity JSON."""
    def record(**kv):
        _PARITY.setdefault(request.node.nodeid, {}).update(_jsonable(kv))
    return record


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    path = os.environ.get("BP_PARITY_JSON") or os.path.join(ROOT, "gpurun_out", "parity_numbers.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):                       # several pytest invocations of one gpurun call share the file
            try:
                old = json.load(open(path)).get("tests", {})
            except Exception:
                old = {}
        old.update(_PARITY)
        json.dump({"written": time.strftime("%Y-%m-%dT%H:%M:%S"), "exitstatus": int(exitstatus),
                   "tolerance": "max|a-ref| / max|ref| per tensor; fp32 bar 1e-4 (north_star), bf16 bar 2e-2",
                   "tests": old}, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass                                           # read-only checkout: the numbers were still printed


@pytest.fixture(scope="session")
def pkg():
    import dnnse_amd
    return dnnse_amd


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O
