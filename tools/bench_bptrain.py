#!/usr/bin/env python3
"""End-to-end rate of the BPtrain-compatible command line on a synthetic Pfile pair (reader + upload + GPU),
C2 geometry: 257-dim frames, 11-frame context, 2827->2048x3->257, ReLU + dropout, bunch 256, traincache 102400.

    python tools/bench_bptrain.py [n_sentences] [frames_per_sentence] [--dp]

Compares  stack=device (raw frames + index tables, windows built on the GPU; default of bptrain)
with      stack=host   (the reference's layout: 11x stacked rows built and uploaded by the host),
each with and without the read-ahead thread.  Prints one JSON line per mode with the rate bptrain logs
("Training pass: ... frames/s").  Synthetic data, random initial weights (the Gen_rand recipe), lrate 0.001."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "bptrain")
HEADER = 32768


def write_pfile_fast(path, sent_lens, data):
    """Same bytes as tests/pfile_util.write_pfile, vectorised."""
    n, d = data.shape
    hdr = ("-pfile_header version 0 size 32768\n-num_sentences %d\n-num_frames %d\n-first_feature_column 2\n"
           "-num_features %d\n-end\n" % (len(sent_lens), n, d)).encode()
    rec = np.empty((n, d + 2), dtype=">u4")
    rec[:, 0] = np.repeat(np.arange(len(sent_lens)), sent_lens)
    rec[:, 1] = np.concatenate([np.arange(l) for l in sent_lens])
    rec[:, 2:] = data.astype(">f4").view(">u4")
    with open(path, "wb") as f:
        f.write(hdr + b"\0" * (HEADER - len(hdr)))
        f.write(rec.tobytes())
        f.write(np.concatenate([[0], np.cumsum(sent_lens)]).astype(">i4").tobytes())


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    nsent = int(pos[0]) if len(pos) > 0 else 500
    flen = int(pos[1]) if len(pos) > 1 else 420
    D, ctx = 257, 11
    ls = [D * ctx, 2048, 2048, 2048, D]
    rs = np.random.default_rng(1)
    lens = [flen] * nsent
    n = sum(lens)
    tmp = tempfile.mkdtemp(prefix="bptrain_bench_")
    fea = rs.standard_normal((n, D), dtype=np.float32)
    write_pfile_fast(os.path.join(tmp, "f.pfile"), lens, fea)
    write_pfile_fast(os.path.join(tmp, "t.pfile"), lens, rs.standard_normal((n, D), dtype=np.float32))
    with open(os.path.join(tmp, "n.norm"), "w") as f:
        f.write("<mean>\n" + "".join("%.9g\n" % v for v in fea.mean(0)) + "<inverse std>\n" +
                "".join("%.9g\n" % v for v in 1.0 / fea.std(0)))
    del fea
    base = ["fea_file=%s/f.pfile" % tmp, "targ_file=%s/t.pfile" % tmp, "norm_file=%s/n.norm" % tmp,
            "train_sent_range=0-%d" % (nsent - 11), "cv_sent_range=%d-%d" % (nsent - 10, nsent - 1),
            "fea_dim=%d" % D, "fea_context=%d" % ctx, "targ_offset=5", "dropoutflag=1", "traincache=102400",
            "bunchsize=256", "gpu_used=1", "init_randem_seed=27863875", "momentum=0.5", "weightcost=0", "lrate=0.001",
            "visible_omit=0.1", "hid_omit=0.2", "layersizes=%s" % ",".join(map(str, ls)),
            "init_randem_weight_max=0.03", "init_randem_weight_min=-0.03", "init_randem_bias_max=0", "init_randem_bias_min=0"]
    # gpu_used=N (ranks share the devices that exist): ONE reader per node -- the shared chunk ring of chunk_ring.h; the
    # node-level rate the reader side sustains is what "frames_per_s" shows once the GPUs stop being the limit
    dp_modes = [("gpu_used=%d one reader per node (bunchsize %d global)" % (g, 256 * g), ["gpu_used=%d" % g, "bunchsize=%d" % (256 * g)])
                for g in (2, 4, 8)] if "--dp" in sys.argv else []
    for mode, extra in [("stack=device prefetch=1", []), ("stack=device prefetch=0", ["prefetch=0"]),
                        ("stack=host prefetch=1", ["stack=host"]), ("stack=host prefetch=0", ["stack=host", "prefetch=0"])] + dp_modes:
        log = os.path.join(tmp, "log")
        t0 = time.time()
        r = subprocess.run([EXE] + base + ["outwts_file=%s/w" % tmp, "log_file=" + log] + extra, capture_output=True, text=True)
        wall = time.time() - t0
        txt = open(log).read()
        m = re.search(r"Training pass: (\d+) samples in ([0-9.]+) s \((\d+) frames/s", txt)
        cv = re.search(r"CV over\. squared error: (\S+)", txt)
        print(json.dumps({"mode": mode, "rc": r.returncode, "train_samples": int(m.group(1)) if m else None,
                          "train_seconds": float(m.group(2)) if m else None, "frames_per_s": int(m.group(3)) if m else None,
                          "cv_sq_err": cv.group(1) if cv else None, "process_wall_s": round(wall, 2)}), flush=True)
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
