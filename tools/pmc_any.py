#!/usr/bin/env python3
"""Mean of every counter per (kernel, grid) from rocprofv3 --pmc csv output dirs: python tools/pmc_any.py out.json dir..."""
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = "%s grid=%s" % (r["Kernel_Name"].split("(")[0][:90], r.get("Grid_Size", "?"))
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: dict({c: sum(v) / len(v) for c, v in cs.items()}, launches=max(len(v) for v in cs.values())) for k, cs in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(out.items()):
    print(k); print("   ", {c: round(x, 1) for c, x in v.items()})
