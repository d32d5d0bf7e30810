"""Test-side restatement of the reference's host data path (Interface.cc:468-1055):
synthetic Pfile writer, libc lrand48/drand48, chunk planner, chunk reader, weight-file writer.  Written
independently of csrc/host/*.cpp to pin that code (the reference's Interface.cc cannot be compiled in
this image, so no reference-generated fixtures exist for it)."""
import struct

import numpy as np

HEADER = 32768


class Rand48(object):
    """glibc srand48 / lrand48 / drand48."""
    A, C, M = 0x5DEECE66D, 0xB, 1 << 48

    def __init__(self, seed):
        self.x = ((seed & 0xFFFFFFFF) << 16) | 0x330E

    def _next(self):
        self.x = (self.A * self.x + self.C) % self.M
        return self.x

    def lrand48(self):
        return self._next() >> 17

    def drand48(self):
        return self._next() / float(self.M)


def rand_index(n, rng):
    """Interface::GetRandIndex (Interface.cc:1044-1055)."""
    v = list(range(n))
    for i in range(n - 1):
        idx = rng.lrand48() % (n - i)
        v[idx], v[n - 1 - i] = v[n - 1 - i], v[idx]
    return v


def write_pfile(path, sent_lens, data):
    """data: [total_frames][dim] float32.  Records (sent_id, frame_id, feat[dim]) big-endian, then the sentence
    table of num_sentences+1 big-endian cumulative offsets."""
    n = int(sum(sent_lens))
    assert data.shape[0] == n
    hdr = ("-pfile_header version 0 size 32768\n-num_sentences %d\n-num_frames %d\n-first_feature_column 2\n"
           "-num_features %d\n-end\n" % (len(sent_lens), n, data.shape[1])).encode()
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * (HEADER - len(hdr)))
        fr = 0
        for s, ln in enumerate(sent_lens):
            for j in range(ln):
                f.write(struct.pack(">ii", s, j))
                f.write(data[fr].astype(">f4").tobytes())
                fr += 1
        f.write(np.concatenate([[0], np.cumsum(sent_lens)]).astype(">i4").tobytes())


def write_norm(path, mean, inv_std):
    with open(path, "w") as f:
        f.write("<mean>\n")
        for v in mean:
            f.write("%.9g\n" % v)
        f.write("<inverse std>\n")
        for v in inv_std:
            f.write("%.9g\n" % v)


def plan(frames_before, total_frames, ctx, cache, st, en):
    cur = 0 if st == 0 else frames_before[st - 1]
    starts, cnt = [cur], 0
    for s in range(st, en + 1):
        inc = frames_before[s] - cur
        cur = frames_before[s]
        cnt += inc - (ctx - 1 if inc >= ctx else inc)
        while cnt >= cache:
            nxt = cur - (cnt - cache)
            if nxt >= total_frames:
                cnt = cache - 1
                break
            starts.append(nxt)
            cnt = cur - nxt - ctx + 1 if cur - nxt > ctx - 1 else 0
    return starts, (len(starts) - 1) * cache + cnt


def read_chunk(fea, tg, sent_of_frame, frames_before, mean, inv_std, starts, total_samples, sent_en, ci, ctx, cache,
               targ_offset, nat, order):
    """fea/tg: raw [frames][dim] float32 as stored; returns (in, targ) with rows placed at order[k]."""
    D = fea.shape[1]
    st = starts[ci]
    if ci == len(starts) - 1:
        need, n = frames_before[sent_en] - st, total_samples - cache * ci
    else:
        need, n = starts[ci + 1] - st, cache
    x = ((fea[st:st + need].astype(np.float32) - mean.astype(np.float32)) * inv_std.astype(np.float32)).astype(np.float32)
    t = tg[st:st + need]
    s0 = D * (ctx + 1) if nat else D * ctx
    out_in = np.zeros((max(n, 0), s0), np.float32)
    out_tg = np.zeros((max(n, 0), tg.shape[1]), np.float32)
    done, cur_frame, k, sent = 0, st, 0, int(sent_of_frame[st])
    while done != need and sent < len(frames_before):
        seg = need - done if frames_before[sent] > need + st else frames_before[sent] - cur_frame
        for j in range(0, seg - ctx + 1):
            if k >= n:
                break
            row = out_in[order[k]]
            row[:D * ctx] = x[done + j:done + j + ctx].reshape(-1)
            if nat:
                s = x[min(done, need - 1)].copy()
                for f in range(1, 6):
                    s = (s + x[min(done + f, need - 1)]).astype(np.float32)
                row[D * ctx:] = (s / np.float32(6.0)).astype(np.float32)
            out_tg[order[k]] = t[min(done + j + targ_offset, need - 1)]
            k += 1
        cur_frame = frames_before[sent]
        sent += 1
        done += seg
    return out_in, out_tg


def write_wts(path, layersizes, W, b):
    """Interface::Writeweights (Interface.cc:411-465): MAT level-4 matrices, type 10."""
    with open(path, "wb") as f:
        for l in range(1, len(layersizes)):
            for name, rows, cols, a in (("weights%d%d" % (l, l + 1), layersizes[l], layersizes[l - 1], W[l]),
                                        ("bias%d" % (l + 1), 1, layersizes[l], b[l])):
                nm = name.encode() + b"\0"
                f.write(struct.pack("<iiiii", 10, rows, cols, 0, len(nm)))
                f.write(nm)
                f.write(np.ascontiguousarray(a, np.float32).tobytes())


def read_wts(path, layersizes):
    raw = open(path, "rb").read()
    o, W, b = 0, [None], [None]
    for l in range(1, len(layersizes)):
        for kind in range(2):
            t, rows, cols, z, nl = struct.unpack_from("<iiiii", raw, o)
            o += 20 + nl
            a = np.frombuffer(raw, np.float32, rows * cols, o).copy()
            o += 4 * rows * cols
            if kind == 0:
                W.append(a.reshape(layersizes[l - 1], layersizes[l]))
            else:
                b.append(a)
    return W, b


def expected_windows(fea, lens, mean_t, istd_t, ctx, nat):
    """Every window of every sentence, in file order, with the noise-aware block from the SENTENCE's first 6 frames
    (sequential fp32 sum / 6.0f, Interface.cc:776-779)."""
    norm = ((fea - mean_t) * istd_t).astype(np.float32)
    rows, o = [], 0
    for ln in lens:
        s = norm[o:o + ln]
        if ln >= ctx:
            nb = None
            if nat:
                acc = s[0].copy()
                for f in range(1, 6):
                    acc = (acc + s[min(f, ln - 1)]).astype(np.float32)
                nb = (acc / np.float32(6.0)).astype(np.float32)
            for j in range(ln - ctx + 1):
                w = s[j:j + ctx].reshape(-1)
                rows.append(np.concatenate([w, nb]) if nat else w)
        o += ln
    return np.stack(rows)
