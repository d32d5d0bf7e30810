"""One rank of a data-parallel run through the in-library exchange (bp_dp_attach): spawned by
tests/test_dp_native.py, one process per rank; ranks may share one device.  No torch import:
the exchange is the library's own (hipIpc peer kernels), the rendezvous a shared-memory block.

    python tests/dp_worker.py <case.json> <rank> <outdir>
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def case_data(c):
    """Inputs every rank (and the checking test) derives identically from the case description."""
    from oracle import bp_numpy as N
    ls = c["ls"]
    W, b = N.glorot_net(ls, seed=c.get("wseed", 5), beta=c.get("beta", 1.0))
    rng = np.random.default_rng(c.get("dseed", 17))
    b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.1 for l in range(1, len(ls))]
    Bg = c["B"] * c["world"]
    n = c["nb"] * Bg + c.get("tail", 0)
    x = rng.normal(size=(n, ls[0])).astype(np.float32)
    t = rng.normal(size=(n, ls[-1])).astype(np.float32)
    return W, b, x, t


def shard_rows(n_frames, global_bunch, world, rank):
    lb = global_bunch // world
    nb = n_frames // global_bunch
    return (np.arange(nb)[:, None] * global_bunch + rank * lb + np.arange(lb)[None, :]).reshape(-1)


def main():
    c = json.load(open(sys.argv[1]))
    rank, outdir = int(sys.argv[2]), sys.argv[3]
    import dnnse_amd
    ls, B, world = c["ls"], c["B"], c["world"]
    W, b, x, t = case_data(c)
    kw = dict(activation=c.get("act", 0), momentum_rule=c.get("rule", 0), compute_dtype=c.get("compute_dtype", 0))
    if c.get("drop"):
        kw.update(dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=99)
    # ranks spread over every visible device (rank r on device r % ndev), so that any box with >= 2 GPUs exercises the
    # exchange ACROSS devices; "ndev": 1 in the case pins them to one device
    ndev = int(c.get("ndev", 0)) or dnnse_amd.device_count()
    g = dnnse_amd.BP_GPU(world, len(ls), ls, B, c.get("lr", 1.0), c.get("m", 0.5), c.get("wc", 0.0), W, b,
                         device=rank % ndev, global_bunchsize=B * world, rank_frame_offset=rank * B,
                         max_chunk_frames=max(4 * B, 64), **kw)
    g.dp_attach(world, rank, c["key"], transport=c.get("transport", 0))
    peers = [g.dp_peer_info(p) for p in range(world)]
    idx = shard_rows(x.shape[0], B * world, world, rank)
    # two calls: the exchange state (epochs, flags) must carry across training calls
    half = (c["nb"] // 2) * B
    if half:
        g.train(half, x[idx[:half]], t[idx[:half]])
    g.train(idx.size - half, x[idx[half:]], t[idx[half:]])
    w, bb = g.get_weights()
    dw, dbb = g.get_deltas()                      # collective: gathers the sharded momentum state
    n_cv = min(x.shape[0], 3 * B + 1)
    cv = g.CrossValid(n_cv, x, t)
    out = {"cv": np.float64(cv), "epochs": np.int64(g.dp_info()[2]), "out": g.forward(x[:n_cv])}
    json.dump({"device": rank % ndev, "ndev": ndev, "peers": peers}, open(os.path.join(outdir, "rank%d.json" % rank), "w"))
    for l in range(1, len(ls)):
        out["W%d" % l], out["b%d" % l], out["dW%d" % l], out["db%d" % l] = w[l], bb[l], dw[l], dbb[l]
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    g.dp_detach()
    g.close()


if __name__ == "__main__":
    main()
