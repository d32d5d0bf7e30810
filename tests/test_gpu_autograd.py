"""GPU parity of the BACKWARD pass against a reference that shares nothing with oracle/ (-m gpu): the gradient the device
computes for one bunch of BASELINE.json's C2 and C3 at their real sizes (bp_grads_resident, through the C ABI) against torch
float64 autograd of the loss written out in tests/torch_ref.py, with the dropout masks drawn by the test's own numpy Philox
(tests/philox_np.py).  Bar: 1e-4 relative per tensor (north_star), bf16 mode 2e-2.
The one effect that can legitimately break 1e-4 at this size is counted, not absorbed: a hidden pre-activation within fp32
rounding of zero gets its ReLU decision from the summation order; such frames (a handful of 1.5 M decisions, each asserted to
lie within 64 rounding units of zero) are removed from BOTH gradients, each side with its own activations."""
import numpy as np
import pytest

from flip_accounting import backprop_rows, relu_flips
from philox_np import drop_mask
from torch_ref import torch_grads
from util import TOL, relerr

pytestmark = pytest.mark.gpu

C2 = [257 * 11, 2048, 2048, 2048, 257]
C3 = [257 * 12, 2048, 2048, 2048, 257]          # 11 frames + the appended noise-estimate block (NAT)


def _mk(pkg, ls, B, W, b, **kw):
    return pkg.BP_GPU(1, len(ls), ls, B, 1.0, 0.5, 0.0, W, b, max_chunk_frames=B, **kw)


@pytest.mark.parametrize("ls,drop", [(C2, True), (C3, False)])
def test_full_size_gradient_matches_torch_float64_autograd(pkg, parity_record, ls, drop):
    pytest.importorskip("torch")
    B, L, seed = 256, len(ls), 31
    W, b = pkg.glorot_net(ls, seed=1, beta=0.5)                      # the bench's init recipe (product code, not oracle/)
    rng = np.random.default_rng(20260927)
    x = rng.standard_normal((B, ls[0]), dtype=np.float32)
    t = rng.standard_normal((B, ls[-1]), dtype=np.float32)
    kw = dict(dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=seed) if drop else {}
    g = _mk(pkg, ls, B, W, b, **kw)
    g.upload_chunk(x, t)
    g.grads_resident(0)                                              # step 0 of a fresh handle: dropout stream position 0
    gw, gb = g.read_grads()
    ys_g = [None] + [g.read_layer_output(l) for l in range(1, L - 1)]
    g.close()
    masks = [drop_mask(seed, 0, l, B, ls[l], 0.1 if l == 0 else 0.2) for l in range(L - 1)] if drop else None
    tw, tb, ys_t, out_t = torch_grads(ls, W, b, x, t, masks)
    ys_g[0] = ys_t[0]                                                # the masked input rows (the device never stores them per bunch)
    if drop:                                                         # the device drew the same hidden masks: dropped units are 0 there
        for l in range(1, L - 1):
            assert not ys_g[l][masks[l] == 1].any(), ("a unit the test's Philox drops is alive on the device", l)
            on_t = ys_t[l] > 0
            assert ((ys_g[l] > 0) == on_t).mean() > 0.9999, l        # (flips are counted below; this catches a wrong mask stream)
    fl = []
    for l in range(1, L - 1):
        fl += [(l,) + f for f in relu_flips(ys_g[l], ys_t[l], ys_g[l - 1], W[l], b[l])]
    print("ReLU decisions that differ from torch float64: %d of %d hidden units: %s"
          % (len(fl), B * sum(ls[1:-1]), [(l, f, n, "%.1e" % m, "%.1e" % sc) for l, f, n, m, sc in fl]))
    assert len(fl) <= 8, fl
    for l, f, n, mag, scale in fl:
        assert mag <= 64.0 * scale, ("a differing ReLU decision that is NOT within rounding of zero", l, f, n, mag, scale)
    rows = sorted(set(f for _, f, _, _, _ in fl))
    if rows:
        keep = np.ones(B, bool); keep[rows] = False
        tw, tb, _, _ = torch_grads(ls, W, b, x, t, masks, keep_rows=keep)          # torch without those frames
        out_g = np.zeros((B, ls[-1]))                                # the device's training-mode output of those frames, from ITS hidden outputs
        out_g[rows] = ys_g[L - 2][rows].astype(np.float64) @ W[L - 1].astype(np.float64) + b[L - 1].astype(np.float64)
        dx_g = backprop_rows(ls, W, ys_g, out_g, t, rows, B)
    worst = {}
    for l in range(1, L):
        Gg, bg = gw[l].astype(np.float64), gb[l].astype(np.float64)
        if rows:
            Gg = Gg - ys_g[l - 1][rows].astype(np.float64).T @ dx_g[l]
            bg = bg - dx_g[l].sum(0)
        worst["W%d" % l], worst["b%d" % l] = relerr(Gg, tw[l]), relerr(bg, tb[l])
    print("gradient vs torch float64 autograd (frames %s removed):" % rows, {k: "%.1e" % v for k, v in worst.items()})
    parity_record(config="C2" if drop else "C3", reference="torch float64 autograd + numpy Philox (nothing from oracle/)",
                  flips=[list(f) for f in fl], gradient_vs_torch_float64=worst, bar="plain 1e-4 with the differing frames removed from both sides")
    assert all(v < TOL for v in worst.values()), worst


@pytest.mark.parametrize("ls,drop", [(C2, True), (C3, False)], ids=["C2", "C3"])
def test_full_size_ten_step_trajectory_matches_torch_float64(pkg, parity_record, ls, drop):
    """The oracle-FREE counterpart of test_gpu_parity.py::test_full_size_ten_steps_small_lrate_plain_bar: C2 / C3 at real size, TEN training
    steps at lrate 0.02 / momentum 0.5 on the device against a trajectory computed here in torch float64 -- autograd gradients of the loss
    written out in tests/torch_ref.py, dropout masks from the test's own numpy Philox, the reference's update rule in float64
    (delta <- m delta - (1-m) lr (G/n + wc W); W <- W + delta: DevFunc.cu:313-318, 270-277) -- nothing under oracle/ is touched.
    Outputs of the trained net on fresh frames and every weight matrix at PLAIN 1e-4.  (Momentum state and the zero-initialised biases carry
    the ReLU-decision effect independently of lrate -- counted and removed exactly in the oracle-based test, recorded raw here.)"""
    pytest.importorskip("torch")
    import torch
    B, L, NS, lr, m, seed = 256, len(ls), 10, 0.02, 0.5, 31
    W, b = pkg.glorot_net(ls, seed=1, beta=0.5)
    rng = np.random.default_rng(20261002)
    x = rng.standard_normal((NS * B, ls[0]), dtype=np.float32)
    t = rng.standard_normal((NS * B, ls[-1]), dtype=np.float32)
    kw = dict(dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=seed) if drop else {}
    g = pkg.BP_GPU(1, L, ls, B, lr, m, 0.0, W, b, max_chunk_frames=NS * B, **kw)
    g.train(NS * B, x, t)                                            # ONE call: ten bunches back to back
    w, bb = g.get_weights()
    dw, dbb = g.get_deltas()
    xf = rng.standard_normal((300, ls[0]), dtype=np.float32)
    out_g = g.forward(xf)
    g.close()
    W64 = [None] + [np.asarray(W[l], np.float64).copy() for l in range(1, L)]
    b64 = [None] + [np.asarray(b[l], np.float64).copy() for l in range(1, L)]
    dW64 = [None] + [np.zeros_like(W64[l]) for l in range(1, L)]
    db64 = [None] + [np.zeros_like(b64[l]) for l in range(1, L)]
    c1 = (1.0 - m) * lr
    for i in range(NS):
        masks = [drop_mask(seed, i, l, B, ls[l], 0.1 if l == 0 else 0.2) for l in range(L - 1)] if drop else None
        gw, gb, _, _ = torch_grads(ls, W64, b64, x[i * B:(i + 1) * B], t[i * B:(i + 1) * B], masks)
        for l in range(1, L):
            dW64[l] = m * dW64[l] - c1 * (gw[l] / B); W64[l] = W64[l] + dW64[l]
            db64[l] = m * db64[l] - c1 * (gb[l] / B); b64[l] = b64[l] + db64[l]
    h = torch.from_numpy(xf.astype(np.float64))                      # CV forward of the trained net (keep-scaled weights, BP_GPU.cu:703-746)
    for l in range(1, L):
        keep = (0.9 if l == 1 else 0.8) if drop else 1.0
        h = keep * (h @ torch.from_numpy(W64[l])) + torch.from_numpy(b64[l])
        if l < L - 1:
            h = torch.clamp(h, min=0.0)
    errs = {"out": relerr(out_g, h.numpy())}
    for l in range(1, L):
        errs["W%d" % l] = relerr(w[l], W64[l]); errs["b%d" % l] = relerr(bb[l], b64[l])
        errs["dW%d" % l] = relerr(dw[l], dW64[l]); errs["db%d" % l] = relerr(dbb[l], db64[l])
    print("ten steps vs the torch float64 trajectory:", {k: "%.1e" % v for k, v in errs.items()})
    parity_record(config="C2" if drop else "C3", lrate=lr, momentum=m, steps=NS, reference="torch float64 autograd trajectory + numpy Philox (nothing from oracle/)",
                  vs_torch_float64=errs, bar="plain 1e-4 on outputs and weight matrices; momentum state and biases recorded (ReLU-decision effect, lrate-independent)")
    assert errs["out"] < TOL, errs
    for l in range(1, L):
        assert errs["W%d" % l] < TOL, (l, errs)


def test_bf16_gradient_matches_an_oracle_free_bf16_reference(pkg, parity_record):
    """compute_dtype = 1 (bf16 GEMM operands, fp32 accumulation; BASELINE.json configs[4]'s arithmetic) at bf16's bar of 2e-2
    against a reference that shares nothing with oracle/: the bunch written out by hand in numpy float64 with bf16 rounding at
    the points where the device stores bf16 (tests/torch_ref.py bf16_grads; pinned on the CPU against the oracle's bf16 mode).
    Rms over each tensor: under bf16 single gradient elements move by a few per cent of the largest when a rounding / ReLU
    boundary falls differently, the tensor as a whole does not.  Against EXACT arithmetic (torch float64 autograd of the same
    loss) the bf16 gradient is 4-6 % off in the lower layers -- measured and printed here, it is what bf16 storage costs,
    not a defect -- so that figure only has to show that the device really computes in bf16."""
    pytest.importorskip("torch")
    from torch_ref import bf16_grads
    ls, B, seed = [2827, 1024, 1024, 257], 256, 5
    W, b = pkg.glorot_net(ls, seed=2, beta=0.5)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((B, ls[0]), dtype=np.float32)
    t = rng.standard_normal((B, ls[-1]), dtype=np.float32)
    g = _mk(pkg, ls, B, W, b, dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=seed, compute_dtype=1)
    g.upload_chunk(x, t)
    g.grads_resident(0)
    gw, gb = g.read_grads()
    g.close()
    masks = [drop_mask(seed, 0, l, B, ls[l], 0.1 if l == 0 else 0.2) for l in range(len(ls) - 1)]
    rw, rb, _, _ = bf16_grads(ls, W, b, x, t, masks)
    tw, tb, _, _ = torch_grads(ls, W, b, x, t, masks)
    rms = lambda a, r: float(np.sqrt(((np.asarray(a, np.float64) - r) ** 2).sum() / max((r ** 2).sum(), 1e-300)))
    worst, exact = {}, {}
    for l in range(1, len(ls)):
        worst["W%d" % l], worst["b%d" % l] = rms(gw[l], rw[l]), rms(gb[l], rb[l])
        exact["W%d" % l] = rms(gw[l], tw[l])
    print("bf16 gradient vs the hand-written bf16 reference (rms):", {k: "%.1e" % v for k, v in worst.items()})
    print("bf16 gradient vs exact arithmetic, torch float64 autograd (rms):", {k: "%.1e" % v for k, v in exact.items()})
    parity_record(gradient_rms_vs_handwritten_bf16_reference=worst, gradient_rms_vs_exact_arithmetic=exact, bar="2e-2 rms")
    assert all(v < 2e-2 for v in worst.values()), worst
    assert all(1e-4 < v < 0.15 for v in exact.values()), exact      # it really is a bf16 computation, and not a broken one
