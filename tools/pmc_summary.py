#!/usr/bin/env python3
"""Summarise rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes into profiles/rNN_pmc_hbm_traffic.json.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> "<note>"

Per (kernel, grid size): number of launches and the median counter value.  Units are KB; FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half).
"""
import csv
import glob
import json
import os
import statistics
import sys


def collect(d, counter):
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            key = "%s grid=%s" % (r["Kernel_Name"].split("(")[0], r.get("Grid_Size", "?"))
            rows.setdefault(key, []).append(float(r["Counter_Value"]))
    return rows


def main():
    fd, wd, out, note = sys.argv[1:5]
    fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    ks = {}
    for k in sorted(set(fe) | set(wr)):
        e = {"launches": len(fe.get(k, wr.get(k, [])))}
        if k in fe:
            m = statistics.median(fe[k])
            e["FETCH_SIZE_KB_median"] = m
            e["fetch_MB_corrected_x2"] = 2 * m * 1024 / 1e6
        if k in wr:
            m = statistics.median(wr[k])
            e["WRITE_SIZE_KB_median"] = m
            e["write_MB"] = m * 1024 / 1e6
        ks[k] = e
    json.dump({"note": note, "kernels": ks}, open(out, "w"), indent=1)
    print("wrote", out, len(ks), "kernels")


if __name__ == "__main__":
    main()
