"""An oracle-INDEPENDENT reference of one training bunch's gradient: torch float64 autograd of the loss the reference
minimises, written out here -- L = (1/B) sum_f sum_d (out[f,d] - t[f,d])^2 (dEdX_L = (2/B)(out - t), DevFunc.cu:253-268,
BP_GPU.cu:588-652) over the frame-wise net with non-inverted dropout (mask applied to the layer OUTPUT, derivative taken
from the post-dropout output, BP_GPU.cu:546-549, DevFunc.cu:81-97).  Nothing under oracle/ is imported: the GPU test that
uses this compares the device with torch, not with this repo's own restatement."""
import numpy as np


def torch_grads(ls, W, b, x, t, masks=None, keep_rows=None, act=0):
    """Gradients dL/dW_l, dL/db_l (lists indexed 1..L-1, float64 numpy) and the hidden outputs ys[l] (post-activation,
    post-dropout; ys[0] = masked input).  masks[l], l = 0..L-2: uint8 [B][ls[l]] with 1 = dropped, or None.
    keep_rows: boolean [B]; frames with False do not enter the loss (their contribution to every gradient is removed) --
    the scale stays 1/B."""
    import torch
    L, B = len(ls), x.shape[0]
    Wt = [None] + [torch.tensor(np.asarray(W[l], np.float64), requires_grad=True) for l in range(1, L)]
    bt = [None] + [torch.tensor(np.asarray(b[l], np.float64), requires_grad=True) for l in range(1, L)]
    h = torch.from_numpy(np.asarray(x, np.float64))
    if masks is not None and masks[0] is not None:
        h = h * torch.from_numpy(1.0 - masks[0].astype(np.float64))
    ys = [h.detach().numpy()]
    for l in range(1, L):
        z = h @ Wt[l] + bt[l]
        if l < L - 1:
            h = torch.clamp(z, min=0.0) if act == 0 else torch.sigmoid(z)
            if masks is not None and masks[l] is not None:
                h = h * torch.from_numpy(1.0 - masks[l].astype(np.float64))
            ys.append(h.detach().numpy())
        else:
            out = z
    d = out - torch.from_numpy(np.asarray(t, np.float64))
    if keep_rows is not None:
        d = d * torch.from_numpy(np.asarray(keep_rows, np.float64))[:, None]
    loss = (d * d).sum() / B
    loss.backward()
    gw = [None] + [Wt[l].grad.numpy() for l in range(1, L)]
    gb = [None] + [bt[l].grad.numpy() for l in range(1, L)]
    return gw, gb, ys, out.detach().numpy()


def bf16_round(a):
    """fp32 -> bf16 (round to nearest even) -> back, as float64 values: the storage rounding of compute_dtype = 1
    (dnn-for-speech-enhancement_amd/csrc/bp_bf16.h f2bf)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def bf16_grads(ls, W, b, x, t, masks=None):
    """The same bunch with bf16 STORAGE of everything a GEMM reads and exact (float64) arithmetic in between -- written
    out by hand (forward, dEdX_L = (2/B)(out - t), dEdX_{l-1} = (y_{l-1} > 0) * dEdX_l . W_l^T, G_l = y_{l-1}^T . dEdX_l),
    because the rounding of the back-propagated errors is not something autograd can express.  Rounded to bf16, as on the
    device: the masked input, every hidden output, every dEdX_l, the weights; NOT rounded: out, the bias, the sums.  ReLU."""
    L, B = len(ls), x.shape[0]
    Wb = [None] + [bf16_round(W[l]) for l in range(1, L)]
    h = np.asarray(x, np.float64)
    if masks is not None and masks[0] is not None:
        h = h * (1.0 - masks[0])
    ys = [bf16_round(h)]
    for l in range(1, L):
        z = ys[l - 1] @ Wb[l] + np.asarray(b[l], np.float64)
        if l < L - 1:
            y = np.maximum(z, 0.0)
            if masks is not None and masks[l] is not None:
                y = y * (1.0 - masks[l])
            ys.append(bf16_round(y))
        else:
            out = z
    dx = {L - 1: bf16_round((2.0 / B) * (out - np.asarray(t, np.float64)))}
    for l in range(L - 1, 1, -1):
        dx[l - 1] = bf16_round((ys[l - 1] > 0) * (dx[l] @ Wb[l].T))
    gw = [None] + [ys[l - 1].T @ dx[l] for l in range(1, L)]
    gb = [None] + [dx[l].sum(0) for l in range(1, L)]
    return gw, gb, ys, out
