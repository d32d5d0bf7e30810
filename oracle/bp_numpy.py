"""Independent numpy fp64 restatement of SURVEY.md Appendix A (one training step) and the CV
forward.  TEST INFRASTRUCTURE ONLY: used to pin oracle/bp_oracle.c and to generate the
fixtures under tests/golden/ (tests/golden/make_golden.py).  PARITY UNPINNED by the reference
(no goldens in /root/reference, reference not buildable here) -- see bp_oracle.c header.

Follows (paths relative to /root/reference): BP_GPU.cu:484-673 (step), :676-773 (CV forward),
DevFunc.cu:34-45,67-97,253-268,313-318 and DevFunc.cu.bak:47-63,275 (Sigmoid / classic rule).
"""
import numpy as np


def act(x, kind):
    if kind == 0:
        return np.where(x > 0, x, 0.0)
    return 1.0 / (1.0 + np.exp(-x))


def dact(y, kind):
    if kind == 0:
        return (y > 0).astype(y.dtype)
    return (1.0 - y) * y


def forward_cv(W, b, x, dropoutflag=0, visible_omit=0.0, hid_omit=0.0, activation=0):
    """W, b: lists indexed 1..L-1 (index 0 unused)."""
    L = len(W)
    y = np.asarray(x, dtype=np.float64)
    for l in range(1, L):
        keep = 1.0
        if dropoutflag == 1:
            keep = (1.0 - visible_omit) if l == 1 else (1.0 - hid_omit)
        xx = y @ (np.asarray(W[l], np.float64) * keep) + np.asarray(b[l], np.float64)
        y = xx if l == L - 1 else act(xx, activation)
    return y


def grads(W, b, x, t, masks=None, activation=0, scale_frames=None):
    """Returns gw, gb (lists 1..L-1), ys (post-dropout layer outputs, ys[0] = masked input), out."""
    L = len(W)
    B = x.shape[0]
    n = B if scale_frames is None else scale_frames
    ys = [np.array(x, dtype=np.float64)]
    for l in range(1, L):
        if masks is not None and masks[l - 1] is not None:
            ys[l - 1] = np.where(np.asarray(masks[l - 1]) != 0, 0.0, ys[l - 1])
        xx = ys[l - 1] @ np.asarray(W[l], np.float64) + np.asarray(b[l], np.float64)
        ys.append(xx if l == L - 1 else act(xx, activation))
    out = ys[L - 1]
    gw, gb = [None] * L, [None] * L
    dedy = None
    for l in range(L - 1, 0, -1):
        if l == L - 1:
            dedx = (2.0 / n) * (out - np.asarray(t, np.float64))
        else:
            dedx = dact(ys[l], activation) * dedy
        if l != 1:
            dedy = dedx @ np.asarray(W[l], np.float64).T
        gw[l] = ys[l - 1].T @ dedx
        gb[l] = dedx.sum(axis=0)
    return gw, gb, ys, out


def update(W, b, dW, db, gw, gb, n, lr, m, wc, momentum_rule=0):
    L = len(W)
    f = 1.0 if momentum_rule == 1 else (1.0 - m)
    for l in range(1, L):
        dW[l] = m * dW[l] - f * lr * (gw[l] / n + wc * W[l])
        W[l] = W[l] + dW[l]
        db[l] = m * db[l] - f * lr * (gb[l] / n)
        b[l] = b[l] + db[l]


def train_steps(W, b, xs, ts, lr, m, wc, masks_per_step=None, activation=0, momentum_rule=0):
    """Run len(xs) steps in fp64 from zero momentum state; returns (W, b, dW, db) fp64 lists."""
    L = len(W)
    W = [None] + [np.array(W[l], np.float64) for l in range(1, L)]
    b = [None] + [np.array(b[l], np.float64) for l in range(1, L)]
    dW = [None] + [np.zeros_like(W[l]) for l in range(1, L)]
    db = [None] + [np.zeros_like(b[l]) for l in range(1, L)]
    for i, (x, t) in enumerate(zip(xs, ts)):
        mk = None if masks_per_step is None else masks_per_step[i]
        gw, gb, _, _ = grads(W, b, x, t, mk, activation)
        update(W, b, dW, db, gw, gb, x.shape[0], lr, m, wc, momentum_rule)
    return W, b, dW, db


def glorot_net(layersizes, seed=1, beta=0.5):
    """Gen_rand_net flag=1 recipe (toolbox/weights/gen_rand_net/Gen_rand_net.cpp:89-101):
    W ~ U(-r, r), r = beta*sqrt(6)/sqrt(prev+cur), bias 0.  (numpy RNG, not libc rand().)"""
    rng = np.random.default_rng(seed)
    L = len(layersizes)
    W, b = [None], [None]
    for l in range(1, L):
        p, c = layersizes[l - 1], layersizes[l]
        r = beta * np.sqrt(6.0) / np.sqrt(p + c)
        W.append(rng.uniform(-r, r, size=(p, c)).astype(np.float32))
        b.append(np.zeros(c, dtype=np.float32))
    return W, b
