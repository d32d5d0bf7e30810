"""Host-side mirror of the reference's `BP_GPU` class (BP_GPU.h:40-88) over the C ABI in
include/bp_c_api.h (ctypes binding of libbp_hip.so, the gfx950 HIP library).

Same constructor argument order, method names, argument meaning and error behaviour as the
reference object so callers/tests read like `BPtrain.cc`:

    obj = BP_GPU(gpu_used, numlayers, layersizes, bunchsize, lrate, momentum, weightcost,
                 weights, bias, dropoutflag, visible_omit, hid_omit)
    obj.train(n_frames, indata, targ)             # BP_GPU.cu:241-331
    err = obj.CrossValid(n_frames, indata, targ)  # BP_GPU.cu:408-479 (SUM of squared errors)
    obj.returnWeights(weights, bias)              # BP_GPU.cu:910-923

There is NO CPU fallback: if the HIP library is missing or no MI355X is visible the
constructor raises.  (The reference prints and calls exit(0) on errors, BP_GPU.cu:20-24;
`strict_exit=True` reproduces that, the default raises BPError so Python callers can react.)
"""
import ctypes as C
import os
import sys

import numpy as np

MAXLAYER = 10          # BP_GPU.h:13
MAXCACHEFRAME = 200000  # BP_GPU.h:14

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libbp_hip.so")

# every symbol include/bp_c_api.h declares
ABI_SYMBOLS = [
    "bp_last_error", "bp_abi_version", "bp_build_target", "bp_create", "bp_destroy", "bp_train_chunk",
    "bp_cv_chunk", "bp_forward", "bp_get_weights", "bp_get_deltas", "bp_upload_chunk",
    "bp_fill_chunk_synthetic", "bp_train_resident", "bp_sync", "bp_grads_resident",
    "bp_grad_layout", "bp_grad_floats", "bp_read_grads", "bp_read_layer_output", "bp_last_train_ms", "bp_time_kernel",
    "bp_upload_chunk_windows", "bp_train_chunk_windows", "bp_cv_chunk_windows",
    "bp_set_hyper", "bp_dp_attach", "bp_dp_attach_ex", "bp_dp_detach", "bp_dp_info", "bp_dp_peer_info", "bp_dp_handoff", "bp_dp_barrier", "bp_dp_allgather",
    "bp_rdv_open", "bp_rdv_barrier", "bp_rdv_allgather", "bp_rdv_close", "bp_device_pci_bus_id", "bp_host_register", "bp_host_unregister",
    "bp_profile_step", "bp_measure_peaks", "bp_device_count", "bp_train_resident_masked", "bp_forward_windows",
]
PROF_KINDS = ["fwd_l1", "fwd_hidden", "fwd_out", "dgrad_out", "dgrad_hidden", "wgrad_update_grouped"]


class BPError(RuntimeError):
    pass


class BPWindowChunk(C.Structure):
    """bp_window_chunk (include/bp_c_api.h): raw frames + index tables, stacked on the device."""
    _fields_ = [
        ("n_samples", C.c_int), ("n_frames", C.c_int), ("fea_dim", C.c_int), ("context", C.c_int), ("n_nat", C.c_int),
        ("fea", C.POINTER(C.c_float)), ("targ_frames", C.POINTER(C.c_float)), ("nat", C.POINTER(C.c_float)),
        ("win_start", C.POINTER(C.c_int)), ("targ_frame", C.POINTER(C.c_int)), ("nat_row", C.POINTER(C.c_int)),
    ]


class BPConfig(C.Structure):
    _fields_ = [
        ("gpu_used", C.c_int), ("numlayers", C.c_int), ("layersizes", C.c_int * MAXLAYER),
        ("bunchsize", C.c_int), ("lrate", C.c_float), ("momentum", C.c_float), ("weightcost", C.c_float),
        ("dropoutflag", C.c_int), ("visible_omit", C.c_float), ("hid_omit", C.c_float),
        ("activation", C.c_int), ("momentum_rule", C.c_int), ("seed", C.c_uint64), ("device", C.c_int),
        ("global_bunchsize", C.c_int), ("rank_frame_offset", C.c_int), ("max_chunk_frames", C.c_int),
        ("compute_dtype", C.c_int),
    ]


_lib = None


def load_library(path=None):
    """dlopen the HIP library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("BP_HIP_LIB", LIB_PATH)   # (BP_HIP_LIB: development builds of the same ABI)
    if not os.path.exists(p):
        raise BPError("HIP library %s not found: build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
    lib = C.CDLL(p)
    fpp = C.POINTER(C.POINTER(C.c_float))
    fp = C.POINTER(C.c_float)
    hp = C.c_void_p
    lib.bp_last_error.restype = C.c_char_p
    lib.bp_build_target.restype = C.c_char_p
    lib.bp_create.argtypes = [C.POINTER(BPConfig), fpp, fpp, C.POINTER(hp)]
    lib.bp_destroy.argtypes = [hp]
    lib.bp_train_chunk.argtypes = [hp, C.c_int, fp, fp]
    lib.bp_cv_chunk.argtypes = [hp, C.c_int, fp, fp, fp]
    lib.bp_forward.argtypes = [hp, C.c_int, fp, fp]
    lib.bp_get_weights.argtypes = [hp, fpp, fpp]
    lib.bp_get_deltas.argtypes = [hp, fpp, fpp]
    lib.bp_upload_chunk.argtypes = [hp, C.c_int, fp, fp]
    lib.bp_upload_chunk_windows.argtypes = [hp, C.POINTER(BPWindowChunk)]
    lib.bp_train_chunk_windows.argtypes = [hp, C.POINTER(BPWindowChunk)]
    lib.bp_cv_chunk_windows.argtypes = [hp, C.POINTER(BPWindowChunk), fp]
    lib.bp_forward_windows.argtypes = [hp, C.POINTER(BPWindowChunk), fp]
    lib.bp_fill_chunk_synthetic.argtypes = [hp, C.c_int, C.c_uint64]
    lib.bp_train_resident.argtypes = [hp, C.c_int, C.c_int]
    lib.bp_sync.argtypes = [hp]
    lib.bp_grads_resident.argtypes = [hp, C.c_int]
    lib.bp_grad_floats.argtypes = [hp, C.POINTER(C.c_size_t)]
    lib.bp_read_grads.argtypes = [hp, fp, C.c_size_t]
    lib.bp_grad_layout.argtypes = [hp, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.bp_last_train_ms.argtypes = [hp, fp, C.POINTER(C.c_int)]
    lib.bp_time_kernel.argtypes = [hp, C.c_int, C.c_int, fp]
    lib.bp_train_resident_masked.argtypes = [hp, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8))]
    lib.bp_profile_step.argtypes = [hp, C.c_int, C.c_int, fp, C.POINTER(C.c_int)]
    lib.bp_measure_peaks.argtypes = [hp, fp, fp]
    lib.bp_set_hyper.argtypes = [hp, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
    lib.bp_dp_attach.argtypes = [hp, C.c_int, C.c_int, C.c_char_p]
    lib.bp_dp_attach_ex.argtypes = [hp, C.c_int, C.c_int, C.c_char_p, C.c_int]
    lib.bp_dp_peer_info.argtypes = [hp, C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bp_dp_handoff.argtypes = [hp, C.POINTER(C.c_int)]
    lib.bp_dp_barrier.argtypes = [hp]
    lib.bp_dp_allgather.argtypes = [hp, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.bp_rdv_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_void_p)]
    lib.bp_rdv_barrier.argtypes = [C.c_void_p]
    lib.bp_rdv_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.bp_rdv_close.argtypes = [C.c_void_p]
    lib.bp_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    lib.bp_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.bp_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.bp_host_unregister.argtypes = [C.c_void_p]
    lib.bp_read_layer_output.argtypes = [hp, C.c_int, fp, C.c_size_t]
    lib.bp_dp_detach.argtypes = [hp]
    lib.bp_dp_info.argtypes = [hp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    if path is None:
        _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ptrs(arrs):
    P = C.POINTER(C.c_float)
    out = (P * MAXLAYER)()
    for i, a in enumerate(arrs):
        if a is not None:
            out[i] = a.ctypes.data_as(P)
    return out


class BP_GPU(object):
    """Drop-in for the reference trainer object; see module docstring."""

    def __init__(self, gpu_used, numlayers, layersizes, bunchsize, lrate, momentum, weightcost, weights, bias,
                 dropoutflag=0, visible_omit=0.0, hid_omit=0.0, activation=0, momentum_rule=0, seed=0, device=0,
                 global_bunchsize=0, rank_frame_offset=0, max_chunk_frames=0, strict_exit=False, compute_dtype=0):
        self._h = None
        self._strict = strict_exit
        self._lib = load_library()
        self.numlayers = int(numlayers)
        self.layersizes = [int(x) for x in list(layersizes)[:numlayers]]
        self.bunchsize = int(bunchsize)
        self.lrate, self.momentum, self.weightcost = float(lrate), float(momentum), float(weightcost)
        self.dropoutflag, self.visible_omit, self.hid_omit = int(dropoutflag), float(visible_omit), float(hid_omit)
        cfg = BPConfig()
        cfg.gpu_used, cfg.numlayers, cfg.bunchsize = int(gpu_used), self.numlayers, self.bunchsize
        for i, s in enumerate(self.layersizes[:MAXLAYER]):
            cfg.layersizes[i] = s
        cfg.lrate, cfg.momentum, cfg.weightcost = self.lrate, self.momentum, self.weightcost
        cfg.dropoutflag, cfg.visible_omit, cfg.hid_omit = self.dropoutflag, self.visible_omit, self.hid_omit
        cfg.activation, cfg.momentum_rule, cfg.seed, cfg.device = int(activation), int(momentum_rule), int(seed), int(device)
        cfg.global_bunchsize, cfg.rank_frame_offset = int(global_bunchsize), int(rank_frame_offset)
        cfg.max_chunk_frames = int(max_chunk_frames)
        cfg.compute_dtype = int(compute_dtype)          # 0 fp32 (reference) | 1 bf16 operands, fp32 accumulate / master weights
        self._cfg = cfg
        if len(self.layersizes) != self.numlayers or self.numlayers < 2 or self.numlayers > MAXLAYER - 1:
            self._fail("numlayers must be in 2..%d and match layersizes" % (MAXLAYER - 1))
        w = [None] * MAXLAYER
        b = [None] * MAXLAYER
        for l in range(1, self.numlayers):
            w[l] = np.ascontiguousarray(weights[l], dtype=np.float32).reshape(-1)
            b[l] = np.ascontiguousarray(bias[l], dtype=np.float32).reshape(-1)
            if w[l].size != self.layersizes[l - 1] * self.layersizes[l] or b[l].size != self.layersizes[l]:
                self._fail("weights[%d]/bias[%d] have the wrong size" % (l, l))
        h = C.c_void_p()
        self._check(self._lib.bp_create(C.byref(cfg), _ptrs(w), _ptrs(b), C.byref(h)))
        self._h = h

    # ------------------------------------------------------------------ errors
    def _fail(self, msg):
        if self._strict:                      # reference convention: printf + exit(0)
            print(msg)
            sys.exit(0)
        raise BPError(msg)

    def _check(self, rc):
        if rc != 0:
            self._fail("%s (status %d)" % (self._lib.bp_last_error().decode(), rc))

    def _in(self, a, n_frames, width, name):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.size < n_frames * width:
            self._fail("%s holds %d floats, need %d" % (name, a.size, n_frames * width))
        return a

    # ------------------------------------------------------------------ reference API
    def _push_hyper(self):
        """The reference reads its public members afresh on every bunch (BP_GPU.cu:488-500): a caller may
        assign obj.lrate / momentum / weightcost / dropoutflag / visible_omit / hid_omit between chunks."""
        self._check(self._lib.bp_set_hyper(self._h, float(self.lrate), float(self.momentum), float(self.weightcost),
                                           int(self.dropoutflag), float(self.visible_omit), float(self.hid_omit)))

    def train(self, n_frames, indata, targ):
        x = self._in(indata, n_frames, self.layersizes[0], "in")
        t = self._in(targ, n_frames, self.layersizes[-1], "targ")
        self._push_hyper()
        self._check(self._lib.bp_train_chunk(self._h, int(n_frames), _fp(x), _fp(t)))

    def CrossValid(self, n_frames, indata, targ):
        x = self._in(indata, n_frames, self.layersizes[0], "in")
        t = self._in(targ, n_frames, self.layersizes[-1], "targ")
        self._push_hyper()
        e = C.c_float(0.0)
        self._check(self._lib.bp_cv_chunk(self._h, int(n_frames), _fp(x), _fp(t), C.byref(e)))
        return float(e.value)

    def returnWeights(self, weights, bias):
        """Fills caller-owned arrays weights[l]/bias[l], l = 1..numlayers-1 (index 0 unused)."""
        for l in range(1, self.numlayers):
            for a, n in ((weights[l], self.layersizes[l - 1] * self.layersizes[l]), (bias[l], self.layersizes[l])):
                if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous and a.size == n):
                    self._fail("returnWeights: arrays must be contiguous float32 of the layer's size")
        self._check(self._lib.bp_get_weights(self._h, _ptrs(weights), _ptrs(bias)))

    # ------------------------------------------------------------------ conveniences / extensions
    def _new_params(self):
        w = [None] + [np.empty((self.layersizes[l - 1], self.layersizes[l]), np.float32) for l in range(1, self.numlayers)]
        b = [None] + [np.empty(self.layersizes[l], np.float32) for l in range(1, self.numlayers)]
        return w, b

    def get_weights(self):
        w, b = self._new_params()
        self.returnWeights(w, b)
        return w, b

    def get_deltas(self):
        w, b = self._new_params()
        self._check(self._lib.bp_get_deltas(self._h, _ptrs(w), _ptrs(b)))
        return w, b

    def forward(self, indata):
        x = np.ascontiguousarray(indata, dtype=np.float32).reshape(-1, self.layersizes[0])
        out = np.empty((x.shape[0], self.layersizes[-1]), np.float32)
        self._check(self._lib.bp_forward(self._h, x.shape[0], _fp(x), _fp(out)))
        return out

    def upload_chunk(self, indata, targ):
        x = np.ascontiguousarray(indata, dtype=np.float32).reshape(-1, self.layersizes[0])
        t = self._in(targ, x.shape[0], self.layersizes[-1], "targ")
        self._check(self._lib.bp_upload_chunk(self._h, x.shape[0], _fp(x), _fp(t)))

    # ---- on-device frame stacking (bp_window_chunk): sample i = fea[win_start[i] : win_start[i]+context] (+ nat[nat_row[i]])
    def _windows(self, fea, targ_frames, context, win_start, targ_frame, nat=None, nat_row=None):
        fea = np.ascontiguousarray(fea, dtype=np.float32)
        if fea.ndim != 2:
            self._fail("windows: fea must be [n_frames][fea_dim]")
        tg = np.ascontiguousarray(targ_frames, dtype=np.float32).reshape(fea.shape[0], self.layersizes[-1])
        ws = np.ascontiguousarray(win_start, dtype=np.int32)
        tf = np.ascontiguousarray(targ_frame, dtype=np.int32)
        c = BPWindowChunk()
        c.n_samples, c.n_frames, c.fea_dim, c.context = int(ws.size), int(fea.shape[0]), int(fea.shape[1]), int(context)
        c.fea, c.targ_frames = _fp(fea), _fp(tg)
        ip = C.POINTER(C.c_int)
        c.win_start, c.targ_frame = ws.ctypes.data_as(ip), tf.ctypes.data_as(ip)
        keep = [fea, tg, ws, tf]
        if nat is not None:
            nat = np.ascontiguousarray(nat, dtype=np.float32).reshape(-1, fea.shape[1])
            nr = np.ascontiguousarray(nat_row, dtype=np.int32)
            c.n_nat, c.nat, c.nat_row = int(nat.shape[0]), _fp(nat), nr.ctypes.data_as(ip)
            keep += [nat, nr]
        return c, keep

    def upload_chunk_windows(self, fea, targ_frames, context, win_start, targ_frame, nat=None, nat_row=None):
        c, keep = self._windows(fea, targ_frames, context, win_start, targ_frame, nat, nat_row)
        self._check(self._lib.bp_upload_chunk_windows(self._h, C.byref(c)))

    def train_windows(self, fea, targ_frames, context, win_start, targ_frame, nat=None, nat_row=None):
        c, keep = self._windows(fea, targ_frames, context, win_start, targ_frame, nat, nat_row)
        self._check(self._lib.bp_train_chunk_windows(self._h, C.byref(c)))

    def CrossValid_windows(self, fea, targ_frames, context, win_start, targ_frame, nat=None, nat_row=None):
        c, keep = self._windows(fea, targ_frames, context, win_start, targ_frame, nat, nat_row)
        e = C.c_float(0.0)
        self._check(self._lib.bp_cv_chunk_windows(self._h, C.byref(c), C.byref(e)))
        return float(e.value)

    def fill_chunk_synthetic(self, n_frames, seed=20260927):
        self._check(self._lib.bp_fill_chunk_synthetic(self._h, int(n_frames), int(seed)))

    def train_resident(self, first_frame, n_frames):
        self._check(self._lib.bp_train_resident(self._h, int(first_frame), int(n_frames)))

    def train_resident_masked(self, first_frame, n_frames, masks):
        """masks[l], l = 0..numlayers-2: uint8 [n_frames][layersizes[l]] (1 = drop) or None -- parity tests only."""
        P = C.POINTER(C.c_uint8)
        arr = (P * MAXLAYER)()
        keep = []
        for l, m in enumerate(masks):
            if m is not None:
                m = np.ascontiguousarray(m, dtype=np.uint8).reshape(int(n_frames), self.layersizes[l])
                keep.append(m)
                arr[l] = m.ctypes.data_as(P)
        self._check(self._lib.bp_train_resident_masked(self._h, int(first_frame), int(n_frames), arr))

    # ---- gradients without the update (parity tests): bp_grads_resident / bp_read_grads / bp_read_layer_output
    def grads_resident(self, first_frame):
        self._check(self._lib.bp_grads_resident(self._h, int(first_frame)))

    def grad_floats(self):
        n = C.c_size_t()
        self._check(self._lib.bp_grad_floats(self._h, C.byref(n)))
        return n.value

    def grad_layout(self, layer):
        o, c = C.c_size_t(), C.c_size_t()
        self._check(self._lib.bp_grad_layout(self._h, int(layer), C.byref(o), C.byref(c)))
        return o.value, c.value

    def read_grads(self):
        """Per-layer weight / bias gradients of the last bp_grads_resident, unpadded: ([None, G_1 [prev][cur], ...], [None, gb_1, ...])."""
        g = np.empty(self.grad_floats(), np.float32)
        self._check(self._lib.bp_read_grads(self._h, _fp(g), g.size))
        pad = lambda v: (v + 63) & ~63
        gw, gb = [None], [None]
        for l in range(1, self.numlayers):
            off, cnt = self.grad_layout(l)
            lp, lc = pad(self.layersizes[l - 1]), pad(self.layersizes[l])
            assert cnt == lp * lc + lc
            gw.append(g[off:off + lp * lc].reshape(lp, lc)[:self.layersizes[l - 1], :self.layersizes[l]].copy())
            gb.append(g[off + lp * lc:off + cnt][:self.layersizes[l]].copy())
        return gw, gb

    def read_layer_output(self, layer):
        y = np.empty((self.bunchsize, self.layersizes[layer]), np.float32)
        self._check(self._lib.bp_read_layer_output(self._h, int(layer), _fp(y), y.size))
        return y

    # ---- in-library data-parallel exchange (bp_dp_attach, include/bp_c_api.h)
    def dp_attach(self, world, rank, key, transport=0):
        """transport: 0 = the library's peer kernels over hipIpc mappings (reduce-scatter by peer reads), 1 = RCCL reduce-scatter /
        all-gather, 2 = the library's peer kernels in push form (every rank writes its slices into the owners' receive buffers)."""
        self._check(self._lib.bp_dp_attach_ex(self._h, int(world), int(rank), str(key).encode(), int(transport)))

    def dp_peer_info(self, peer):
        """(device ordinal in the peer's process, PCI bus id, transport, acquire mode) of rank `peer`."""
        dev, tr, aq = C.c_int(), C.c_int(), C.c_int()
        buf = C.create_string_buffer(32)
        self._check(self._lib.bp_dp_peer_info(self._h, int(peer), C.byref(dev), buf, 32, C.byref(tr), C.byref(aq)))
        return int(dev.value), buf.value.decode(), int(tr.value), int(aq.value)

    def dp_handoff(self):
        """True: gradient segments are handed to the exchange inside the running weight-gradient launch (tile counters);
        False: event + kernel boundary per group of layers."""
        v = C.c_int()
        self._check(self._lib.bp_dp_handoff(self._h, C.byref(v)))
        return bool(v.value)

    def dp_barrier(self):
        self._check(self._lib.bp_dp_barrier(self._h))

    def dp_allgather_f64(self, value, world):
        mine = (C.c_double * 1)(float(value))
        out = (C.c_double * int(world))()
        self._check(self._lib.bp_dp_allgather(self._h, mine, 8, out))
        return [float(v) for v in out]

    def dp_detach(self):
        self._check(self._lib.bp_dp_detach(self._h))

    def dp_info(self):
        w, r, n = C.c_int(), C.c_int(), C.c_uint()
        self._check(self._lib.bp_dp_info(self._h, C.byref(w), C.byref(r), C.byref(n)))
        return int(w.value), int(r.value), int(n.value)

    def sync(self):
        self._check(self._lib.bp_sync(self._h))

    def last_train_ms(self):
        ms, nb = C.c_float(), C.c_int()
        self._check(self._lib.bp_last_train_ms(self._h, C.byref(ms), C.byref(nb)))
        return float(ms.value), int(nb.value)

    def time_kernel(self, which, iters=50):
        ms = C.c_float()
        self._check(self._lib.bp_time_kernel(self._h, int(which), int(iters), C.byref(ms)))
        return float(ms.value)

    def profile_step(self, first_frame, n_bunches):
        """{class: (avg ms per launch inside the step, launches per step)} -- bp_profile_step."""
        ms = (C.c_float * len(PROF_KINDS))()
        cnt = (C.c_int * len(PROF_KINDS))()
        self._check(self._lib.bp_profile_step(self._h, int(first_frame), int(n_bunches), ms, cnt))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(PROF_KINDS)}

    def measure_peaks(self):
        a, b = C.c_float(), C.c_float()
        self._check(self._lib.bp_measure_peaks(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def close(self):
        if self._h is not None:
            self._lib.bp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int()
    lib = load_library()
    if lib.bp_device_count(C.byref(n)) != 0:
        raise BPError(lib.bp_last_error().decode())
    return int(n.value)


def device_pci_bus_id(device):
    lib = load_library()
    buf = C.create_string_buffer(32)
    if lib.bp_device_pci_bus_id(int(device), buf, 32) != 0:
        raise BPError(lib.bp_last_error().decode())
    return buf.value.decode()


class Rendezvous(object):
    """bp_rdv_* (include/bp_c_api.h): the host-only rendezvous of the data-parallel ranks; works without a GPU."""

    def __init__(self, key, world, rank, timeout_s=30.0):
        self._lib = load_library()
        self.world, self.rank = int(world), int(rank)
        self._r = C.c_void_p()
        if self._lib.bp_rdv_open(str(key).encode(), self.world, self.rank, float(timeout_s), C.byref(self._r)) != 0:
            self._r = None
            raise BPError(self._lib.bp_last_error().decode())

    def barrier(self):
        if self._lib.bp_rdv_barrier(self._r) != 0:
            raise BPError(self._lib.bp_last_error().decode())

    def allgather_f64(self, value):
        mine = (C.c_double * 1)(float(value))
        out = (C.c_double * self.world)()
        if self._lib.bp_rdv_allgather(self._r, mine, 8, out) != 0:
            raise BPError(self._lib.bp_last_error().decode())
        return [float(v) for v in out]

    def close(self):
        if self._r is not None:
            self._lib.bp_rdv_close(self._r)
            self._r = None
