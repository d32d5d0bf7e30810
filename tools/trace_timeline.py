#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 --kernel-trace csv: start offset, duration, gap to the
previous kernel end on the same queue, queue id, short kernel name.   python tools/trace_timeline.py <dir> [N]"""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = {}
for r in rows:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].split("(")[0]
    name = name.replace("bp_gemm_multi<GemmKernel", "multi").replace("bp_gemm", "gemm")[:70]
    print("%9.1f us  dur %7.1f  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name))
