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
