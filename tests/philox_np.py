"""Philox4x32-10 and the dropout-mask keying of the HIP path (dnn-for-speech-enhancement_amd/csrc/bp_kernels.h, drop_words4 /
bp_mask_input), restated in numpy for the tests: a third implementation next to the device's and the oracle's, so that a
GPU parity test can build its reference without importing anything under oracle/.
Keying: counter = (idx_lo, idx_hi, layer, step), key = (seed_lo, seed_hi), idx = (global_frame >> 2) * width + unit;
the unit's word is word[global_frame & 3]; it is dropped iff word < uint32(p * 2^32)  (reference: cuRAND uniform < p,
BP_GPU.cu:534-551 -- a different generator, seeded from time(NULL))."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3; k0, k1 python ints.  Returns the 4 output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & MASK32, p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (h1 ^ c1 ^ np.uint64(k0)) & MASK32, l1, (h0 ^ c3 ^ np.uint64(k1)) & MASK32, l0
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def drop_threshold(p):
    t = float(p) * 4294967296.0
    return 0 if t <= 0 else (4294967295 if t >= 4294967295.0 else int(t))


def drop_mask(seed, step, layer, n_frames, width, p, frame_off=0):
    """uint8 [n_frames][width], 1 = dropped: the mask of `layer`'s output (layer 0 = the input frames) in training step
    `step` for bunch rows 0..n_frames-1 whose global frame index is row + frame_off."""
    thr = drop_threshold(p)
    if thr == 0:
        return np.zeros((n_frames, width), np.uint8)
    gf = np.arange(n_frames, dtype=np.uint64)[:, None] + np.uint64(frame_off)
    idx = (gf >> np.uint64(2)) * np.uint64(width) + np.arange(width, dtype=np.uint64)[None, :]
    w = philox4x32_10(idx & MASK32, idx >> np.uint64(32), np.full(idx.shape, layer, np.uint64), np.full(idx.shape, step, np.uint64),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    sel = (gf & np.uint64(3)).astype(np.int64) + np.zeros(idx.shape, np.int64)
    word = np.choose(sel, w)
    return (word < np.uint32(thr)).astype(np.uint8)
