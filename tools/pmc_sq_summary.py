#!/usr/bin/env python3
"""Counter-based MFMA utilisation per kernel from a tools/pmc_any.py summary of the SQ passes:

    python tools/pmc_sq_summary.py <sq.json> <out.json> <sources_sha> "<note>"

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): SIMD-cycles with an MFMA in flight over the SIMD-cycles
of CUs that were busy.  rocprofv3's derived `MfmaUtil` (kept beside it when the pass collected it) divides by GRBM_GUI_ACTIVE
instead, which under PMC's serialised dispatch includes the idle time between a kernel's dispatch and its first wave -- it
understates short kernels.  profiles/r05_mfma_util.json is this tool's output; bench.py reads it."""
import json
import sys


def main():
    src, out, sha, note = sys.argv[1:5]
    sq = json.load(open(src))
    ks = {}
    for k, v in sq.items():
        busy, cu = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("SQ_BUSY_CU_CYCLES")
        if not busy or not cu:
            continue
        e = {"mfma_busy_frac": busy / (4.0 * cu), "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CU_CYCLES": cu, "launches": v.get("launches")}
        if "MfmaUtil" in v:
            e["rocprofv3_MfmaUtil_percent"] = v["MfmaUtil"]
        if "GRBM_GUI_ACTIVE" in v:
            e["GRBM_GUI_ACTIVE"] = v["GRBM_GUI_ACTIVE"]
        ks[k] = e
    json.dump({"note": note, "sources_sha": sha, "kernels": ks}, open(out, "w"), indent=1)
    for k, e in sorted(ks.items()):
        print("%-100s mfma_busy_frac %.3f%s" % (k[:100], e["mfma_busy_frac"], "   MfmaUtil %.1f %%" % e["rocprofv3_MfmaUtil_percent"] if "rocprofv3_MfmaUtil_percent" in e else ""))


if __name__ == "__main__":
    main()
