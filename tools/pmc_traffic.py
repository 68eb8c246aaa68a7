#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one run each, as MI355X_MICROARCH.md 'HBM'
prescribes) into per-kernel HBM bytes per launch.

    python tools/pmc_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies the
128-B read requests at 64 B, so it is doubled (the guide calibrates this for 16-B-per-lane streaming reads;
8-B-per-lane reads, as these kernels issue, are stated as uncalibrated there: the figure is indicative)."""
import csv, glob, json, os, sys
from collections import defaultdict

CLASS = {"k_linearize": "k_linearize", "k_schur": "k_schur", "k_backsub": "k_backsub", "k_cost": "k_cost",
         "k_solve_small": "pcg", "k_solve_components": "pcg", "k_gather_S": "k_gather_S",
         "k_bseg_factor": "k_bseg_factor", "k_banded_factor": "k_banded_factor", "k_bseg_assemble": "k_bseg_assemble"}


def per_kernel(d, counter):
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            for k, cls in CLASS.items():
                if k + "<" in name or name.startswith(k + "(") or (" " + k + "(") in name or k in name.split("(")[0]:
                    vals[cls].append(float(r["Counter_Value"]))
                    break
    acc = {}
    for cls, v in vals.items():  # launches enqueued after the solver terminated exit at once: leave them out
        keep = [x for x in v if x >= 0.05 * max(v)]
        acc[cls] = [sum(keep), len(keep)]
    return acc


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {}
    for cls in sorted(set(fe) | set(wr)):
        fe.setdefault(cls, [0.0, 0]); wr.setdefault(cls, [0.0, 0])
        f = fe[cls][0] / max(fe[cls][1], 1) * 1024.0
        w = wr[cls][0] / max(wr[cls][1], 1) * 1024.0
        res[cls] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "traffic_bytes": 2 * f + w,
                    "dispatches": [fe[cls][1], wr[cls][1]]}
    json.dump({"unit": "bytes per launch (mean over the dispatches that did work; launches that exited at once are left out)",
               "correction": "FETCH_SIZE x 1024 x 2 (gfx950), WRITE_SIZE x 1024", "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
