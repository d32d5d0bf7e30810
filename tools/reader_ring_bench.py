#!/usr/bin/env python3
"""Node-level reader of `bptrain gpu_used=N` at configs[3] geometry (257-dim frames, 11-frame context, 102400-sample chunks, global
minibatch 2048) on the CPU: synthetic Pfile pair -> tools/bin/reader_ring_bench with 1, 2, 4, 8 forked ranks.
    python tools/reader_ring_bench.py [n_sentences] [frames_per_sentence] [passes]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_bptrain import write_pfile_fast  # noqa: E402

EXE = os.path.join(ROOT, "tools", "bin", "reader_ring_bench")


def main():
    nsent = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    flen = int(sys.argv[2]) if len(sys.argv) > 2 else 420
    passes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    D, ctx = 257, 11
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "reader_ring_bench.cc"),
                           os.path.join(ROOT, "dnn-for-speech-enhancement_amd", "csrc", "host", "pfile_reader.cpp"), "-o", EXE])
    rs = np.random.default_rng(1)
    lens = [flen] * nsent
    n = sum(lens)
    tmp = tempfile.mkdtemp(prefix="ring_bench_", dir="/tmp")
    fea = rs.standard_normal((n, D), dtype=np.float32)
    write_pfile_fast(os.path.join(tmp, "f.pfile"), lens, fea)
    write_pfile_fast(os.path.join(tmp, "t.pfile"), lens, rs.standard_normal((n, D), dtype=np.float32))
    with open(os.path.join(tmp, "n.norm"), "w") as f:
        f.write("<mean>\n" + "".join("%.9g\n" % v for v in fea.mean(0)) + "<inverse std>\n" + "".join("%.9g\n" % v for v in 1.0 / fea.std(0)))
    del fea
    print("# %d sentences x %d frames = %d raw frames of %d floats (+ as many target frames), files in the page cache; %d online CPUs (nproc), "
          "affinity %d" % (nsent, flen, n, D, os.cpu_count(), len(os.sched_getaffinity(0))), flush=True)
    for f in ("f.pfile", "t.pfile"):
        open(os.path.join(tmp, f), "rb").read()                   # page cache
    for drain in ("tables",) + (("copy",) if "--copy" in sys.argv else ()):
        for world in (1, 2, 4, 8):
            subprocess.check_call([EXE, tmp + "/f.pfile", tmp + "/t.pfile", tmp + "/n.norm", str(D), str(ctx), "5", str(D), "102400", str(D * ctx),
                                   "0", str(nsent - 1), "1", str(world), "2048", drain, str(passes)])
            sys.stdout.flush()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
