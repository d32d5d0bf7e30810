#!/bin/bash
# per-chunk hand-over times of a 16-chunk bptrain run (C2 geometry): is the end-to-end rate below the GPU-alone rate because of the first chunks?
cd "$(dirname "$0")/.."
export BPTRAIN_CHUNK_TIMES=1
python - <<'P'
import subprocess, sys, os, re
sys.argv = ["bench_bptrain.py", "4000", "420"]
src = open("tools/bench_bptrain.py").read()
# run only the first mode and show bptrain's stderr
src = src.replace('("stack=device prefetch=0", ["prefetch=0"]),', '').replace('("stack=host prefetch=1", ["stack=host"]), ("stack=host prefetch=0", ["stack=host", "prefetch=0"])] + dp_modes', '] + dp_modes')
src = src.replace('print(json.dumps({"mode": mode', 'print(r.stderr[-1500:]); print(json.dumps({"mode": mode')
exec(compile(src, "bench_bptrain_times", "exec"), {"__file__": os.path.abspath("tools/bench_bptrain.py"), "__name__": "__main__"})
P
