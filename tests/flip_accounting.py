"""ReLU-decision accounting shared by the full-size GPU parity tests (plain numpy, nothing from oracle/).
A hidden pre-activation within fp32 rounding of zero gets its ReLU decision from the GEMM's summation order; the frame it
belongs to then contributes differently to whole gradient columns.  These helpers find such units and compute, in fp64 and
from ONE side's own activations, what the affected frames contribute to every layer's gradient."""
import numpy as np


def relu_flips(y_a, y_ref, y_prev, Wl, bl):
    """Units of one hidden layer whose ReLU state differs between two sides although both saw the same inputs and (up to
    rounding) the same pre-activation.  Returns [(frame, unit, |x| of the side that is on, rounding scale)], rounding scale
    = 2^-24 * (sum_k |y_prev[f,k] * W[k,n]| + |b[n]|): the size of one fp32 rounding error of that dot product.
    (Units dropped by dropout are 0 on both sides and never differ.)"""
    fl = []
    for f, n in zip(*np.nonzero((y_a > 0) != (y_ref > 0))):
        mag = float(max(abs(y_a[f, n]), abs(y_ref[f, n])))
        scale = float((np.abs(y_prev[f].astype(np.float64) * Wl[:, n].astype(np.float64)).sum() + abs(float(bl[n]))) * 2.0 ** -24)
        fl.append((int(f), int(n), mag, scale))
    return fl


def backprop_rows(ls, W, ys, out, t, rows, n_scale):
    """fp64 dEdX_l rows of the given frames from that side's OWN activations (dEdX_L = (2/n)(out - t), BP_GPU.cu:630;
    dEdX_{l-1} = (y_{l-1} > 0) * dEdX_l . W_l^T, :611-637): what those frames contribute to every layer's gradient."""
    L = len(ls)
    dx = {L - 1: (2.0 / n_scale) * (out[rows].astype(np.float64) - t[rows].astype(np.float64))}
    for l in range(L - 1, 1, -1):
        dx[l - 1] = (ys[l - 1][rows] > 0) * (dx[l] @ W[l].astype(np.float64).T)
    return dx


# The bar of a full-size tensor after a ReLU decision HAS differed (counted by the caller) at lrate 1: both the device and the
# fp32 restatement of the reference are then correct fp32 trajectories on opposite sides of a discontinuity, and the yardstick
# is the fp64-accumulated trajectory:  |gpu - fp64| <= K_FP64 * |fp32 oracle - fp64|,  directly, no additive term -- the device
# may be at most K_FP64 times as far from exact arithmetic as the reference-order fp32 arithmetic is.
# K_FP64 = 2 (round 5: 4 plus an additive 1e-4*max).  What the data of round 6 supports (profiles/r06_parity_numbers.json):
# single device, C2 / C3, 2 steps at lrate 1: the ratio is 6e-6 ... 1.6e-3 on every tensor -- the device's MFMA accumulation
# (blocked, two chains) makes the SAME decisions as fp64 accumulation, it is the reference's sequential fp32 order that flips;
# 8 ranks x 256 frames (12.6 M hidden units per global bunch, several flips per bunch on EACH side): 0.31 ... 1.24 over the six
# tensors that miss the plain bar.  There each side's distance is the largest of a few independent one-frame contributions of
# the same size distribution, so a ratio of order 1 is what two equally good fp32 orders give; 2 leaves 1.6x over the largest
# ratio seen and is the smallest round factor that does.
K_FP64 = 2.0


def fp64_bounded(dist_gpu, dist_fp32_oracle, max_fp64=None, tol=None):
    return dist_gpu <= K_FP64 * dist_fp32_oracle
