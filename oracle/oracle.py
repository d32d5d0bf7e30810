"""ctypes wrapper around oracle/libbp_oracle.so (bp_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of bp_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the product package.
PARITY UNPINNED by the reference (it has no tests/goldens and cannot be built here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbp_oracle.so")
MAXLAYER = 10


class OracleCfg(C.Structure):
    _fields_ = [
        ("numlayers", C.c_int),
        ("layersizes", C.c_int * MAXLAYER),
        ("lrate", C.c_float),
        ("momentum", C.c_float),
        ("weightcost", C.c_float),
        ("dropoutflag", C.c_int),
        ("visible_omit", C.c_float),
        ("hid_omit", C.c_float),
        ("activation", C.c_int),
        ("momentum_rule", C.c_int),
        ("acc_double", C.c_int),
        ("seed", C.c_uint64),
        ("compute_dtype", C.c_int),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "bp_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        fpp = C.POINTER(C.POINTER(C.c_float))
        upp = C.POINTER(C.POINTER(C.c_uint8))
        cfgp = C.POINTER(OracleCfg)
        fp = C.POINTER(C.c_float)
        _lib.oracle_forward.argtypes = [cfgp, fpp, fpp, C.c_int, fp, fp]
        _lib.oracle_forward.restype = None
        _lib.oracle_crossvalid.argtypes = [cfgp, fpp, fpp, C.c_int, C.c_int, fp, fp]
        _lib.oracle_crossvalid.restype = C.c_float
        _lib.oracle_grads.argtypes = [cfgp, fpp, fpp, C.c_int, fp, fp, upp, C.c_int, fpp, fpp, fp, fp]
        _lib.oracle_grads.restype = None
        _lib.oracle_update.argtypes = [cfgp, fpp, fpp, fpp, fpp, fpp, fpp, C.c_int]
        _lib.oracle_update.restype = None
        _lib.oracle_train_bunch.argtypes = [cfgp, fpp, fpp, fpp, fpp, C.c_int, fp, fp, upp, C.c_int,
                                            C.c_uint32, C.c_uint64]
        _lib.oracle_train_bunch.restype = None
        _lib.oracle_train_chunk.argtypes = [cfgp, fpp, fpp, fpp, fpp, C.c_int, C.c_int, fp, fp, C.c_int,
                                            C.POINTER(C.c_uint32)]
        _lib.oracle_train_chunk.restype = C.c_int
        _lib.oracle_fill_mask.argtypes = [cfgp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int,
                                          C.POINTER(C.c_uint8)]
        _lib.oracle_fill_mask.restype = None
        _lib.oracle_act_floats.argtypes = [cfgp, C.c_int]
        _lib.oracle_act_floats.restype = C.c_size_t
        _lib.bp_drop_threshold.argtypes = [C.c_float]
        _lib.bp_drop_threshold.restype = C.c_uint32
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ptr_array(arrs, ctype=C.c_float):
    """arrs: list indexed like the reference (index 0 unused -> NULL)."""
    P = C.POINTER(ctype)
    out = (P * MAXLAYER)()
    for i, a in enumerate(arrs):
        if a is not None:
            out[i] = a.ctypes.data_as(P)
    return out


class Oracle:
    """Stateful mirror of BP_GPU on the CPU (weights/bias/delta owned here, fp32 numpy)."""

    def __init__(self, layersizes, bunchsize, lrate=1.0, momentum=0.5, weightcost=0.0,
                 weights=None, bias=None, dropoutflag=0, visible_omit=0.0, hid_omit=0.0,
                 activation=0, momentum_rule=0, acc_double=False, seed=0, compute_dtype=0):
        L = len(layersizes)
        assert 2 <= L <= MAXLAYER - 1
        self.layersizes = list(layersizes)
        self.L = L
        self.bunchsize = bunchsize
        self.cfg = OracleCfg()
        self.cfg.numlayers = L
        for i, s in enumerate(layersizes):
            self.cfg.layersizes[i] = s
        self.cfg.lrate, self.cfg.momentum, self.cfg.weightcost = lrate, momentum, weightcost
        self.cfg.dropoutflag = dropoutflag
        self.cfg.visible_omit, self.cfg.hid_omit = visible_omit, hid_omit
        self.cfg.activation, self.cfg.momentum_rule = activation, momentum_rule
        self.cfg.acc_double = 1 if acc_double else 0
        self.cfg.seed = seed
        self.cfg.compute_dtype = int(compute_dtype)      # 1 = bf16 operands (BASELINE configs[4])
        self.W = [None] + [np.array(weights[l], dtype=np.float32, order="C").reshape(
            layersizes[l - 1], layersizes[l]).copy() for l in range(1, L)]
        self.b = [None] + [np.array(bias[l], dtype=np.float32).reshape(layersizes[l]).copy()
                           for l in range(1, L)]
        self.dW = [None] + [np.zeros_like(self.W[l]) for l in range(1, L)]
        self.db = [None] + [np.zeros_like(self.b[l]) for l in range(1, L)]
        self.step = 0

    # -- helpers
    def _pp(self, arrs):
        return _ptr_array(arrs)

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        B = x.shape[0]
        out = np.empty((B, self.layersizes[-1]), dtype=np.float32)
        lib().oracle_forward(C.byref(self.cfg), self._pp(self.W), self._pp(self.b), B, _fp(x), _fp(out))
        return out

    def crossvalid(self, x, t):
        x = np.ascontiguousarray(x, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        return float(lib().oracle_crossvalid(C.byref(self.cfg), self._pp(self.W), self._pp(self.b),
                                             self.bunchsize, x.shape[0], _fp(x), _fp(t)))

    def grads(self, x, t, masks=None, scale_frames=None):
        """Returns (gw, gb, acts, out); x is copied (the C code masks it in place)."""
        x = np.array(x, dtype=np.float32, order="C")
        t = np.ascontiguousarray(t, dtype=np.float32)
        B = x.shape[0]
        gw = [None] + [np.empty_like(self.W[l]) for l in range(1, self.L)]
        gb = [None] + [np.empty_like(self.b[l]) for l in range(1, self.L)]
        nact = lib().oracle_act_floats(C.byref(self.cfg), B)
        acts = np.empty(nact, dtype=np.float32)
        out = np.empty((B, self.layersizes[-1]), dtype=np.float32)
        mk = None
        if masks is not None:
            ms = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
            mk = _ptr_array(ms, C.c_uint8)
            self._keep = ms
        lib().oracle_grads(C.byref(self.cfg), self._pp(self.W), self._pp(self.b), B, _fp(x), _fp(t), mk,
                           B if scale_frames is None else scale_frames, self._pp(gw), self._pp(gb),
                           _fp(acts), _fp(out))
        ys, o = [x], 0
        for l in range(1, self.L):
            n = B * self.layersizes[l]
            ys.append(acts[o:o + n].reshape(B, self.layersizes[l]))
            o += n
        return gw, gb, ys, out

    def update(self, gw, gb, n):
        lib().oracle_update(C.byref(self.cfg), self._pp(self.W), self._pp(self.b), self._pp(self.dW),
                            self._pp(self.db), self._pp(gw), self._pp(gb), n)

    def fill_mask(self, step, layer, B, gframe0=0):
        w = self.layersizes[layer]
        m = np.empty((B, w), dtype=np.uint8)
        lib().oracle_fill_mask(C.byref(self.cfg), step, layer, gframe0, B, w,
                               m.ctypes.data_as(C.POINTER(C.c_uint8)))
        return m

    def train_bunch(self, x, t, masks=None, gen_masks=True, gframe0=0):
        x = np.array(x, dtype=np.float32, order="C")
        t = np.ascontiguousarray(t, dtype=np.float32)
        mk = None
        if masks is not None:
            ms = [None if m is None else np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
            mk = _ptr_array(ms, C.c_uint8)
        lib().oracle_train_bunch(C.byref(self.cfg), self._pp(self.W), self._pp(self.b), self._pp(self.dW),
                                 self._pp(self.db), x.shape[0], _fp(x), _fp(t), mk, 1 if gen_masks else 0,
                                 self.step, gframe0)
        self.step += 1

    def train(self, x, t, gen_masks=True):
        """BP_GPU::train semantics (partial last bunch dropped). Returns #bunches."""
        x = np.array(x, dtype=np.float32, order="C")
        t = np.ascontiguousarray(t, dtype=np.float32)
        st = C.c_uint32(self.step)
        n = lib().oracle_train_chunk(C.byref(self.cfg), self._pp(self.W), self._pp(self.b), self._pp(self.dW),
                                     self._pp(self.db), self.bunchsize, x.shape[0], _fp(x), _fp(t),
                                     1 if gen_masks else 0, C.byref(st))
        self.step = st.value
        return n


def set_threads(n):
    lib().oracle_set_threads(int(n))


def max_threads():
    return int(lib().oracle_max_threads())


def drop_threshold(p):
    return int(lib().bp_drop_threshold(p))
