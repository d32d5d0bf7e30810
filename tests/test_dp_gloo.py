"""CPU test of the data-parallel driver (dnn-for-speech-enhancement_amd/dp.py) with a 2-rank
`gloo` process group: the exchange logic (shard -> local gradients -> all-reduce(SUM) -> identical
update with n = global bunch) is exercised with the oracle plugged in as the per-rank engine (the
product engine is the HIP library; the oracle is used here only as the checker's compute)."""
import os
import socket

import numpy as np
import pytest

from util import relerr

from oracle import bp_numpy as N

LS, BG, WORLD, NB = [24, 16, 10, 6], 16, 2, 3


class OracleEngine(object):
    """Test double with the HipEngine interface: grads(first_frame) -> flat tensor, update()."""

    def __init__(self, O, torch, W, b, x, t, world):
        self.torch = torch
        self.o = O.Oracle(LS, BG, 1.0, 0.5, 0.001, W, b)
        self.x, self.t, self.world = x, t, world
        self.lb = BG // world
        self.sizes = [(LS[l - 1] * LS[l], LS[l]) for l in range(1, len(LS))]
        self.flat = torch.zeros(sum(a + c for a, c in self.sizes), dtype=torch.float32)

    def grads(self, first_frame):
        sl = slice(first_frame, first_frame + self.lb)
        gw, gb, _, _ = self.o.grads(self.x[sl], self.t[sl], scale_frames=BG)
        parts = []
        for l in range(1, len(LS)):
            parts += [gw[l].reshape(-1), gb[l].reshape(-1)]
        self.flat.copy_(self.torch.from_numpy(np.concatenate(parts)))
        return self.flat

    # layer-by-layer interface used by dp_step_overlapped
    nlayers = len(LS)

    def forward(self, first_frame):
        self.grads(first_frame)                   # the oracle computes every layer at once
        self._stash = self.flat.clone()
        self.flat.zero_()

    def backward_layer(self, l):
        off = sum(a + c for a, c in self.sizes[:l - 1])
        cnt = sum(self.sizes[l - 1])
        self.flat[off:off + cnt] = self._stash[off:off + cnt]
        return self.flat[off:off + cnt]

    def update_layer(self, l):
        if self._mode == "pipe":                  # keep this layer's summed segment; apply when all are in
            off = sum(a + c for a, c in self.sizes[:l - 1]); cnt = sum(self.sizes[l - 1])
            self._summed[off:off + cnt] = self.flat[off:off + cnt]
            if l == len(LS) - 1:
                self.flat.copy_(self._summed)
                self.update()

    def advance(self):
        if self._mode != "pipe":
            self.update()

    # finer split used by DPPipeline
    _mode = "layer"

    def forward_layer(self, first_frame, l):
        self._mode = "pipe"
        self._first = first_frame
        if not hasattr(self, "_summed"):
            self._summed = self.torch.zeros_like(self.flat)

    def dgrads(self):
        self.grads(self._first)                   # every update of the previous bunch has been applied
        self._stash = self.flat.clone()

    def wgrad_layer(self, l):
        return self.backward_layer(l)

    def update(self):
        g = self.flat.numpy()
        gw, gb, o = [None], [None], 0
        for l in range(1, len(LS)):
            a, c = self.sizes[l - 1]
            gw.append(np.ascontiguousarray(g[o:o + a]).reshape(LS[l - 1], LS[l])); o += a
            gb.append(np.ascontiguousarray(g[o:o + c])); o += c
        self.o.update(gw, gb, BG)


def _worker(rank, port, q, overlapped):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import dnnse_amd  # noqa: F401  (registers the package so its dp module can be imported)
    from importlib import import_module
    dp = import_module("dnn_for_speech_enhancement_amd.dp")
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    W, b = N.glorot_net(LS, seed=3, beta=2.0)
    rng = np.random.default_rng(12)
    x = rng.normal(size=(NB * BG + 5, LS[0])).astype(np.float32)      # trailing partial bunch
    t = rng.normal(size=(NB * BG + 5, LS[-1])).astype(np.float32)
    rows = dp.shard_rows(x.shape[0], BG, WORLD, rank)                  # this rank's slice of every bunch
    eng = OracleEngine(O, torch, W, b, x[rows], t[rows], WORLD)
    lb = BG // WORLD
    if overlapped == "pipeline":
        pipe = dp.DPPipeline(eng, dist)
        for i in range(NB):
            pipe.step(i * lb)
        pipe.flush()
    else:
        for i in range(NB):
            (dp.dp_step_overlapped if overlapped else dp.dp_step)(eng, dist, i * lb)
    q.put((rank, [w.copy() for w in eng.o.W[1:]], [v.copy() for v in eng.o.b[1:]], rows[:4].tolist(), len(rows)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlapped", [False, True, "pipeline"])
def test_two_rank_gloo_matches_single_device(oracle_mod, overlapped):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, port, q, overlapped)) for r in range(WORLD)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(WORLD)], key=lambda r: r[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards: rank r gets rows [i*Bg + r*Bg/G, ...) of every FULL global bunch
    assert res[0][3] == [0, 1, 2, 3] and res[1][3] == [8, 9, 10, 11] and res[0][4] == NB * BG // WORLD
    # every rank ends with bit-identical state
    for a, c in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, c)
    # and it equals single-device training at the same global bunch up to summation order
    W, b = N.glorot_net(LS, seed=3, beta=2.0)
    rng = np.random.default_rng(12)
    x = rng.normal(size=(NB * BG + 5, LS[0])).astype(np.float32)
    t = rng.normal(size=(NB * BG + 5, LS[-1])).astype(np.float32)
    o = oracle_mod.Oracle(LS, BG, 1.0, 0.5, 0.001, W, b)
    assert o.train(x, t) == NB
    for l in range(1, len(LS)):
        assert relerr(res[0][1][l - 1], o.W[l]) < 1e-5
        assert relerr(res[0][2][l - 1], o.b[l]) < 1e-5
