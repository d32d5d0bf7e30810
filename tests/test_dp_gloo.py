"""CPU test (world_size 2, `gloo`) of the data-parallel SEMANTICS the in-library exchange implements (SURVEY.md 8e;
donor: the reference's commented-out train_bunch_multi, BP_GPU.cu:775-908): every rank computes the gradients of its
shard of each global minibatch with dEdX scaled by 2/global, ONE sum over ranks, the identical update with
n = global bunch -- and the result equals single-device training at the global bunch up to summation order.  The
per-rank compute here is the oracle (the checker's arithmetic; the product's exchange runs on the GPU through hipIpc
and is covered by tests/test_dp_native.py); what this test pins is the sharding rule (which rows a rank owns, partial
last minibatch dropped, BP_GPU.cu:315-318) and the scaling rules, with a real multi-process all-reduce."""
import os
import socket

import numpy as np
import pytest

from util import relerr

from oracle import bp_numpy as N

LS, BG, WORLD, NB = [24, 16, 10, 6], 16, 2, 3


def shard_rows(n_frames, global_bunch, world, rank):
    """Rows of `rank`: its Bg/world frames of every FULL global minibatch (same rule as bptrain's shard_rows and
    tests/dp_worker.py)."""
    lb, nb = global_bunch // world, n_frames // global_bunch
    return (np.arange(nb)[:, None] * global_bunch + rank * lb + np.arange(lb)[None, :]).reshape(-1)


def _data():
    W, b = N.glorot_net(LS, seed=3, beta=2.0)
    rng = np.random.default_rng(12)
    x = rng.normal(size=(NB * BG + 5, LS[0])).astype(np.float32)      # trailing partial minibatch
    t = rng.normal(size=(NB * BG + 5, LS[-1])).astype(np.float32)
    return W, b, x, t


def _worker(rank, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    W, b, x, t = _data()
    rows = shard_rows(x.shape[0], BG, WORLD, rank)
    o = O.Oracle(LS, BG, 1.0, 0.5, 0.001, W, b)
    lb = BG // WORLD
    for i in range(NB):
        sl = rows[i * lb:(i + 1) * lb]
        gw, gb, _, _ = o.grads(x[sl], t[sl], scale_frames=BG)          # dEdX_L = (2/Bg)(out - t) on the local rows
        flat = torch.from_numpy(np.concatenate([np.concatenate([gw[l].reshape(-1), gb[l]]) for l in range(1, len(LS))]))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)                    # the one exchange of the minibatch
        g, off = flat.numpy(), 0
        for l in range(1, len(LS)):
            nw = LS[l - 1] * LS[l]
            gw[l] = np.ascontiguousarray(g[off:off + nw]).reshape(LS[l - 1], LS[l]); off += nw
            gb[l] = np.ascontiguousarray(g[off:off + LS[l]]); off += LS[l]
        o.update(gw, gb, BG)                                           # n = GLOBAL bunch
    q.put((rank, [w.copy() for w in o.W[1:]], [v.copy() for v in o.b[1:]], rows[:4].tolist(), len(rows)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_device(oracle_mod):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(WORLD)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(WORLD)], key=lambda r: r[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards: rank r gets rows [i*Bg + r*Bg/G, ...) of every FULL global minibatch
    assert res[0][3] == [0, 1, 2, 3] and res[1][3] == [8, 9, 10, 11] and res[0][4] == NB * BG // WORLD
    for a, c in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):     # every rank ends with bit-identical state
        assert np.array_equal(a, c)
    W, b, x, t = _data()
    o = oracle_mod.Oracle(LS, BG, 1.0, 0.5, 0.001, W, b)               # single device, same global bunch
    assert o.train(x, t) == NB
    for l in range(1, len(LS)):
        assert relerr(res[0][1][l - 1], o.W[l]) < 1e-5
        assert relerr(res[0][2][l - 1], o.b[l]) < 1e-5
