#!/usr/bin/env python3
"""Step time of the bf16 compute mode (compute_dtype=1) next to fp32 on the C2 shape and on the per-GPU shape of
BASELINE.json configs[4] (2827 -> 4096 x 5 -> 257, 512 frames).  Not the benchmark line (bench.py is fp32 C2); the
bf16 path is a parity configuration whose kernel is not tuned to the bf16 MFMA peak.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dnnse_amd  # noqa: E402


def run(name, ls, B, dtype, steps=100, drop=True):
    W, b = dnnse_amd.glorot_net(ls, seed=1, beta=0.5)
    kw = dict(dropoutflag=1, visible_omit=0.1, hid_omit=0.2, seed=1) if drop else {}
    chunk = B * 50
    g = dnnse_amd.BP_GPU(1, len(ls), ls, B, 0.01, 0.5, 0.0, W, b, max_chunk_frames=chunk, compute_dtype=dtype, **kw)
    g.fill_chunk_synthetic(chunk, 7)
    g.train_resident(0, chunk)            # warm-up
    g.sync()
    ms, nb = 0.0, 0
    for _ in range(steps // 50):
        g.train_resident(0, chunk)
        m, k = g.last_train_ms()
        ms += m; nb += k
    flops = (6 * sum(ls[i - 1] * ls[i] for i in range(1, len(ls))) - 2 * ls[0] * ls[1]) * B
    print(json.dumps({"case": name, "dtype": "bf16" if dtype else "f32", "frames_per_step": B, "ms_per_step": ms / nb,
                      "frames_per_s": B * nb / (ms * 1e-3), "TFLOPs": flops * nb / (ms * 1e-3) / 1e12}), flush=True)
    g.close()


if __name__ == "__main__":
    C2 = [2827, 2048, 2048, 2048, 257]
    C5 = [2827, 4096, 4096, 4096, 4096, 4096, 257]
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # e.g. "c2bf16" to run a single case (profiling)
    for dt in (0, 1):
        if only in ("", "c2" + ("bf16" if dt else "f32")):
            run("C2 3x2048, 256 frames", C2, 256, dt)
    for dt in (0, 1):
        if only in ("", "c5" + ("bf16" if dt else "f32")):
            run("configs[4] per-GPU shape 5x4096, 512 frames", C5, 512, dt, drop=False)
