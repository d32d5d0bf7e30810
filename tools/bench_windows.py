#!/usr/bin/env python3
"""Device side of SURVEY 8f N3 at the C2 geometry: a 51200-sample chunk trained from raw frames + index tables
(train_windows: every bunch stacks and masks its own rows, bp_stage_bunch) against the same chunk handed over stacked
by the host (train).  Wall-clock per chunk including the upload, and frames/s.  One JSON line per mode.
    python tools/bench_windows.py          (rocprofv3 --kernel-trace --stats on it gives bp_stage_bunch's duration)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dnnse_amd  # noqa: E402

D, ctx, B, n = 257, 11, 256, 51200
ls = [D * ctx, 2048, 2048, 2048, D]
rs = np.random.default_rng(3)
W, b = dnnse_amd.glorot_net(ls, seed=1, beta=0.5)
n_frames = n + 4000
fea = rs.standard_normal((n_frames, D), dtype=np.float32)
tg = rs.standard_normal((n_frames, D), dtype=np.float32)
ws = rs.integers(0, n_frames - ctx + 1, size=n).astype(np.int32)
tf = (ws + ctx // 2).astype(np.int32)
for mode in ("windows", "stacked"):
    g = dnnse_amd.BP_GPU(1, len(ls), ls, B, 0.001, 0.5, 0.0, W, b, max_chunk_frames=n, dropoutflag=1, visible_omit=0.1,
                         hid_omit=0.2, seed=1)
    if mode == "stacked":
        idx = ws[:, None] + np.arange(ctx)[None, :]
        rows = np.ascontiguousarray(fea[idx].reshape(n, ctx * D))
        trows = np.ascontiguousarray(tg[tf])
    times = []
    for rep in range(4):
        t0 = time.perf_counter()
        if mode == "windows":
            g.train_windows(fea, tg, ctx, ws, tf)
        else:
            g.train(n, rows, trows)
        g.sync()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    print(json.dumps({"mode": mode, "chunk_samples": n, "s_per_chunk_incl_upload": best, "frames_per_s": n / best,
                      "ms_per_bunch": best / (n // B) * 1e3}), flush=True)
    g.close()
