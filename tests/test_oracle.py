"""CPU tests (-m "not gpu"): pin the C oracle (oracle/bp_oracle.c) against the committed
golden fixtures (numpy fp64 restatement), against torch autograd of the equivalent loss, and
check the host-visible pieces of its dropout stream."""
import numpy as np
import pytest

from util import TOL, golden_cases, load_golden, relerr

from oracle import bp_numpy as N


@pytest.mark.parametrize("name", golden_cases())
@pytest.mark.parametrize("acc_double", [False, True])
def test_oracle_matches_golden(oracle_mod, name, acc_double):
    c = load_golden(name)
    o = oracle_mod.Oracle(c["ls"], c["B"], lrate=c["lr"], momentum=c["m"], weightcost=c["wc"], weights=c["W"],
                          bias=c["b"], activation=c["act"], momentum_rule=c["rule"], acc_double=acc_double,
                          dropoutflag=1 if c["has_drop"] else 0, visible_omit=c["drop"][0], hid_omit=c["drop"][1])
    gw, gb, ys, out = o.grads(c["xs"][0], c["ts"][0], None if c["masks"] is None else c["masks"][0])
    assert relerr(out, c["out0"]) < 1e-5
    for l in range(1, c["L"]):
        if c["gw"][l] is not None:
            assert relerr(gw[l], c["gw"][l]) < 1e-5
        assert relerr(gb[l], c["gb"][l]) < 1e-5
    for s in range(c["steps"]):
        o.train_bunch(c["xs"][s], c["ts"][s], masks=None if c["masks"] is None else c["masks"][s], gen_masks=False)
    for l in range(1, c["L"]):
        assert relerr(o.W[l], c["Wf"][l]) < 1e-5
        assert relerr(o.b[l], c["bf"][l]) < 1e-5
    # CV forward with keep-scaled weights + summed squared error (BP_GPU.cu:408-479)
    o2 = oracle_mod.Oracle(c["ls"], max(1, c["B"] // 2 + 1), weights=c["W"], bias=c["b"], activation=c["act"],
                           dropoutflag=1 if c["has_drop"] else 0, visible_omit=c["drop"][0], hid_omit=c["drop"][1])
    assert relerr(o2.forward(c["xs"][0]), c["cv_out"]) < 1e-5
    assert abs(o2.crossvalid(c["xs"][0], c["ts"][0]) - c["cv_sqerr"]) < 1e-4 * abs(c["cv_sqerr"])


@pytest.mark.parametrize("act", [0, 1])
def test_oracle_gradient_is_autograd_of_mse(oracle_mod, act):
    """Steps 4+6 of SURVEY App. A are the exact gradient of (1/B) sum (out-t)^2."""
    torch = pytest.importorskip("torch")
    ls, B = [20, 16, 9, 6], 10
    W, b = N.glorot_net(ls, seed=11, beta=2.0)
    rng = np.random.default_rng(3)
    b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.2 for l in range(1, len(ls))]
    x = rng.normal(size=(B, ls[0])).astype(np.float32)
    t = rng.normal(size=(B, ls[-1])).astype(np.float32)
    o = oracle_mod.Oracle(ls, B, weights=W, bias=b, activation=act, acc_double=True)
    gw, gb, _, out = o.grads(x, t)
    tw = [None] + [torch.tensor(W[l], dtype=torch.float64, requires_grad=True) for l in range(1, len(ls))]
    tb = [None] + [torch.tensor(b[l], dtype=torch.float64, requires_grad=True) for l in range(1, len(ls))]
    y = torch.tensor(x, dtype=torch.float64)
    for l in range(1, len(ls)):
        y = y @ tw[l] + tb[l]
        if l != len(ls) - 1:
            y = torch.relu(y) if act == 0 else torch.sigmoid(y)
    loss = ((y - torch.tensor(t, dtype=torch.float64)) ** 2).sum() / B
    loss.backward()
    assert relerr(out, y.detach().numpy()) < 1e-6
    for l in range(1, len(ls)):
        assert relerr(gw[l], tw[l].grad.numpy()) < 1e-5
        assert relerr(gb[l], tb[l].grad.numpy()) < 1e-5


def test_partial_last_bunch_is_dropped(oracle_mod):
    """BP_GPU.cu:315-318: train() ignores a trailing partial bunch; CV keeps it (:450-453)."""
    ls, B = [6, 5, 4], 4
    W, b = N.glorot_net(ls, seed=2, beta=1.0)
    rng = np.random.default_rng(0)
    x = rng.normal(size=(11, 6)).astype(np.float32)
    t = rng.normal(size=(11, 4)).astype(np.float32)
    a = oracle_mod.Oracle(ls, B, weights=W, bias=b)
    c = oracle_mod.Oracle(ls, B, weights=W, bias=b)
    assert a.train(x, t) == 2
    assert c.train(x[:8], t[:8]) == 2
    for l in (1, 2):
        assert np.array_equal(a.W[l], c.W[l])
    full = oracle_mod.Oracle(ls, B, weights=W, bias=b)
    assert abs(full.crossvalid(x, t) - ((full.forward(x).astype(np.float64) - t) ** 2).sum()) < 1e-3


def test_philox_mask_statistics_and_keying(oracle_mod):
    """u in (0,1], P(drop) = p; masks differ per step/layer, and are a function of the GLOBAL
    frame index (data-parallel invariance)."""
    ls = [64, 48, 8]
    W, b = N.glorot_net(ls, seed=1)
    o = oracle_mod.Oracle(ls, 256, weights=W, bias=b, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=1234)
    m0 = o.fill_mask(0, 0, 256)
    m1 = o.fill_mask(0, 1, 256)
    assert abs(m0.mean() - 0.1) < 0.01 and abs(m1.mean() - 0.2) < 0.015
    assert not np.array_equal(m0, o.fill_mask(1, 0, 256))
    # shard [64,128) of the global bunch == rows 64..127 of the full mask
    assert np.array_equal(o.fill_mask(0, 0, 64, gframe0=64), m0[64:128])
    assert oracle_mod.drop_threshold(0.0) == 0 and oracle_mod.drop_threshold(1.0) == 0xFFFFFFFF
    assert oracle_mod.drop_threshold(0.5) == 0x80000000


def test_dp_shards_sum_to_single_device(oracle_mod):
    """Gradient of the global bunch == sum of shard gradients computed with scale 2/B_global
    (the semantics the RCCL path implements)."""
    ls, Bg, G = [10, 12, 5], 16, 4
    W, b = N.glorot_net(ls, seed=4, beta=2.0)
    rng = np.random.default_rng(8)
    x = rng.normal(size=(Bg, 10)).astype(np.float32)
    t = rng.normal(size=(Bg, 5)).astype(np.float32)
    o = oracle_mod.Oracle(ls, Bg, weights=W, bias=b, acc_double=True)
    gw, gb, _, _ = o.grads(x, t)
    sw = [None, 0, 0]
    sb = [None, 0, 0]
    for r in range(G):
        sl = slice(r * Bg // G, (r + 1) * Bg // G)
        w_, b_, _, _ = o.grads(x[sl], t[sl], scale_frames=Bg)
        for l in (1, 2):
            sw[l] = sw[l] + w_[l].astype(np.float64)
            sb[l] = sb[l] + b_[l].astype(np.float64)
    for l in (1, 2):
        assert relerr(sw[l], gw[l]) < 1e-6 and relerr(sb[l], gb[l]) < 1e-6


@pytest.mark.parametrize("act,rule,wc,drop", [(0, 0, 0.0, True), (0, 1, 0.001, False), (1, 0, 0.01, True), (1, 1, 0.0, False)])
def test_oracle_trajectory_matches_torch_fp64_at_1024_wide(oracle_mod, act, rule, wc, drop):
    """Independent pin of the oracle beyond one gradient (the reference's CUDA path cannot run here and holds no goldens):
    a multi-step TRAJECTORY on 1024-wide layers -- momentum, weight cost, both momentum rules, both activations, injected
    dropout masks, the double 1/B (DevFunc.cu:263,317), then the CV forward with keep-scaled weights -- against torch
    float64 autograd of (1/B) sum (out-t)^2 plus the update rule written out in tensor algebra."""
    torch = pytest.importorskip("torch")
    ls, B, steps, lr, m = [300, 1024, 1024, 64], 48, 3, 0.7, 0.6
    pv, ph = 0.15, 0.25
    W, b = N.glorot_net(ls, seed=21, beta=1.0)
    rng = np.random.default_rng(5)
    b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.1 for l in range(1, len(ls))]
    L = len(ls)
    o = oracle_mod.Oracle(ls, B, lr, m, wc, W, b, activation=act, momentum_rule=rule, acc_double=True,
                          dropoutflag=1 if drop else 0, visible_omit=pv, hid_omit=ph)
    tw = [None] + [torch.tensor(W[l], dtype=torch.float64) for l in range(1, L)]
    tb = [None] + [torch.tensor(b[l], dtype=torch.float64) for l in range(1, L)]
    dw = [None] + [torch.zeros_like(tw[l]) for l in range(1, L)]
    db = [None] + [torch.zeros_like(tb[l]) for l in range(1, L)]
    c1 = lr if rule == 1 else (1.0 - m) * lr
    for s in range(steps):
        x = rng.normal(size=(B, ls[0])).astype(np.float32)
        t = rng.normal(size=(B, ls[-1])).astype(np.float32)
        masks = None
        if drop:
            masks = [(rng.random(size=(B, ls[l])) < (pv if l == 0 else ph)).astype(np.uint8) for l in range(L - 1)]
        o.train_bunch(x, t, masks=masks, gen_masks=False)
        ws = [None] + [tw[l].clone().requires_grad_(True) for l in range(1, L)]
        bs = [None] + [tb[l].clone().requires_grad_(True) for l in range(1, L)]
        y = torch.tensor(x, dtype=torch.float64)
        for l in range(1, L):
            if masks is not None:
                y = y * torch.tensor(1 - masks[l - 1].astype(np.float64))          # non-inverted dropout of the layer INPUT
            y = y @ ws[l] + bs[l]
            if l != L - 1:
                y = torch.relu(y) if act == 0 else torch.sigmoid(y)
        (((y - torch.tensor(t, dtype=torch.float64)) ** 2).sum() / B).backward()
        for l in range(1, L):
            dw[l] = m * dw[l] - c1 * (ws[l].grad / B + wc * tw[l]); tw[l] = tw[l] + dw[l]
            db[l] = m * db[l] - c1 * (bs[l].grad / B); tb[l] = tb[l] + db[l]
    for l in range(1, L):
        assert relerr(o.W[l], tw[l].numpy()) < 2e-6, ("W", l, relerr(o.W[l], tw[l].numpy()))       # fp32 storage of W / delta
        assert relerr(o.dW[l], dw[l].numpy()) < 2e-5, ("dW", l, relerr(o.dW[l], dw[l].numpy()))
        assert relerr(o.b[l], tb[l].numpy()) < 2e-5 and relerr(o.db[l], db[l].numpy()) < 2e-5, ("b", l)
    # CV forward: weights scaled by keep when dropout is configured (BP_GPU.cu:726-746), bias not scaled
    x = rng.normal(size=(B + 5, ls[0])).astype(np.float32)
    t = rng.normal(size=(B + 5, ls[-1])).astype(np.float32)
    y = torch.tensor(x, dtype=torch.float64)
    for l in range(1, L):
        keep = (1.0 - (pv if l == 1 else ph)) if drop else 1.0
        y = y @ (tw[l] * keep) + tb[l]
        if l != L - 1:
            y = torch.relu(y) if act == 0 else torch.sigmoid(y)
    assert relerr(o.forward(x), y.numpy()) < 1e-5
    sq = float(((y - torch.tensor(t, dtype=torch.float64)) ** 2).sum())
    assert abs(o.crossvalid(x, t) - sq) < 1e-4 * sq


def test_test_side_philox_equals_oracle_mask_stream(oracle_mod):
    """tests/philox_np.py (the numpy Philox the oracle-free GPU backward test draws its masks with) produces the oracle's
    masks: seeds with and without high bits, several steps, layers, and frame offsets that are not multiples of 4."""
    from philox_np import drop_mask
    ls, B = [37, 50, 21, 9], 23
    W, b = N.glorot_net(ls, seed=1)
    for seed in (31, (5 << 32) | 7):
        o = oracle_mod.Oracle(ls, B, 1.0, 0.5, 0.0, W, b, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=seed)
        for step in (0, 3):
            for layer in (0, 1, 2):
                for off in (0, 5, 8):
                    assert np.array_equal(o.fill_mask(step, layer, B, off), drop_mask(seed, step, layer, B, ls[layer], 0.1 if layer == 0 else 0.2, off))


@pytest.mark.parametrize("act", [0, 1])
def test_torch_reference_of_the_gpu_backward_test_equals_oracle_gradient(oracle_mod, act):
    """tests/torch_ref.py (torch float64 autograd with non-inverted dropout on the layer outputs, optional removal of
    frames from the loss) against the oracle's gradient with the same masks: the reference the GPU backward test uses is
    itself pinned on the CPU, and the oracle once more by something that shares no code with it."""
    pytest.importorskip("torch")
    from torch_ref import torch_grads
    ls, B = [70, 96, 64, 33], 48
    W, b = N.glorot_net(ls, seed=4, beta=1.5)
    rng = np.random.default_rng(8)
    b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.2 for l in range(1, len(ls))]
    x = rng.normal(size=(B, ls[0])).astype(np.float32)
    t = rng.normal(size=(B, ls[-1])).astype(np.float32)
    o = oracle_mod.Oracle(ls, B, weights=W, bias=b, activation=act, acc_double=True, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=9)
    masks = [o.fill_mask(0, l, B) for l in range(len(ls) - 1)]
    gw, gb, ys, out = o.grads(x, t, masks=masks)
    tw, tb, tys, tout = torch_grads(ls, W, b, x, t, masks, act=act)
    assert relerr(out, tout) < 1e-6
    for l in range(1, len(ls)):
        assert relerr(gw[l], tw[l]) < 1e-5 and relerr(gb[l], tb[l]) < 1e-5, l
    for l in range(len(ls) - 1):
        assert relerr(ys[l], tys[l]) < 1e-6
    # frames removed from the loss == the full gradient minus those frames' contributions
    keep = np.ones(B, bool); keep[[3, 17]] = False
    kw, kb, _, _ = torch_grads(ls, W, b, x, t, masks, keep_rows=keep, act=act)
    xs, ts, ms = x[[3, 17]], t[[3, 17]], [m[[3, 17]] for m in masks]
    rw, rb, _, _ = torch_grads(ls, W, b, xs, ts, ms, act=act)
    for l in range(1, len(ls)):
        assert relerr(kw[l], tw[l] - rw[l] * (2.0 / B)) < 1e-9, l


def test_hand_written_bf16_reference_equals_the_oracle_bf16_mode(oracle_mod):
    """tests/torch_ref.py bf16_grads (numpy float64 with bf16 storage rounding, the reference of the oracle-free bf16 GPU
    test) against the oracle's compute_dtype = 1 gradient with the same masks."""
    from torch_ref import bf16_grads, bf16_round
    assert np.array_equal(bf16_round(np.array([1.0, 1.00390625, 1.01171875, -3.140625], np.float32)), [1.0, 1.0, 1.015625, -3.140625])
    ls, B = [70, 96, 64, 33], 48
    W, b = N.glorot_net(ls, seed=4, beta=1.5)
    rng = np.random.default_rng(8)
    b = [None] + [rng.normal(size=ls[l]).astype(np.float32) * 0.2 for l in range(1, len(ls))]
    x = rng.normal(size=(B, ls[0])).astype(np.float32)
    t = rng.normal(size=(B, ls[-1])).astype(np.float32)
    o = oracle_mod.Oracle(ls, B, weights=W, bias=b, compute_dtype=1, acc_double=True, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=9)
    masks = [o.fill_mask(0, l, B) for l in range(len(ls) - 1)]
    gw, gb, ys, out = o.grads(x, t, masks=masks)
    rw, rb, rys, rout = bf16_grads(ls, W, b, x, t, masks)
    assert relerr(out, rout) < 1e-5
    for l in range(1, len(ls)):
        assert relerr(gw[l], rw[l]) < 2e-3 and relerr(gb[l], rb[l]) < 2e-3, (l, relerr(gw[l], rw[l]), relerr(gb[l], rb[l]))
